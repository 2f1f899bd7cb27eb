"""CPU oracle (test infrastructure). See oracle/fq_oracle.py."""

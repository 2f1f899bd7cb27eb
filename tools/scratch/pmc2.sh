cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA --output-format csv -d /tmp/pm -o p -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
python - <<'PY'
import csv, collections
acc=collections.defaultdict(list)
for row in csv.DictReader(open('/tmp/pm/p_counter_collection.csv')):
    if 'fq_kron64' in row['Kernel_Name']: acc[row['Counter_Name']].append(float(row['Counter_Value']))
for k,v in acc.items(): print(k, sum(v[5:])/len(v[5:]))
PY

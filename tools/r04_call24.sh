#!/bin/bash
# round 4, GPU call 24: tiles kernel with the bank rotation for CPR = 14 / 18 / 22: parity, timing, LDS conflict counter
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; unset FQHIP_LIB
O=gpurun_out/r04c26; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kron_tiles.py -x -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 300 python tools/time_kron.py 80 112 16384 packed f16 128 144 8192 packed f16 128 144 8192 packedr f16 128 144 8192 packed bf16 168 176 8192 packed f16 144 192 8192 packed f16 86 128 16384 packed f16 2>&1 | grep -v amdgpu.ids > $O/time.txt; cat $O/time.txt
mkdir -p $O/pmc; cd /tmp
CMD="python $GRAFT_REPO_ROOT/tools/run_op.py kron128x144 30"
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc/p2 -o p -- $CMD > $GRAFT_REPO_ROOT/$O/pmc/p2.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc/p1 -o p -- $CMD > $GRAFT_REPO_ROOT/$O/pmc/p1.log 2>&1
cd $GRAFT_REPO_ROOT
for p in p1 p2; do f=$(find $O/pmc/$p -name "*counter_collection.csv" | head -1); echo "== $p"; [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'tiles' in r.get('Kernel_Name', ''):
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(acc.items()):
    print(f"  fq_kron_tiles_kernel {k:28s} avg={sum(v)/len(v):16.1f} n={len(v)}")
PY
done > $O/pmc_128x144.txt 2>&1
cat $O/pmc_128x144.txt
rm -rf $O/pmc

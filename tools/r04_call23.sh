#!/bin/bash
# round 4, GPU call 23: bits != 4 quantiser parity; 168x176 on the one-group tiles config; PMC of 128x144 (logs kept)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; unset FQHIP_LIB
O=gpurun_out/r04c23; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_quant_bits.py tests/test_gpu_quant.py tests/test_gpu_kron_tiles.py -x -q > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
timeout 200 python tools/time_kron.py 168 176 8192 packed f16 168 176 8192 packedr f16 168 176 8192 packed bf16 2>&1 | grep -v amdgpu.ids > $O/time_168.txt; cat $O/time_168.txt
mkdir -p $O/pmc; cd /tmp
CMD="python $GRAFT_REPO_ROOT/tools/run_op.py kron128x144 30"
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc/p2 -o p -- $CMD > $GRAFT_REPO_ROOT/$O/pmc/p2.log 2>&1; echo rc=$?
tail -5 $GRAFT_REPO_ROOT/$O/pmc/p2.log | cut -c1-300
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc/p1 -o p -- $CMD > $GRAFT_REPO_ROOT/$O/pmc/p1.log 2>&1; echo rc=$?
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc/p3 -o p -- $CMD > $GRAFT_REPO_ROOT/$O/pmc/p3.log 2>&1; echo rc=$?
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc/p4 -o p -- $CMD > $GRAFT_REPO_ROOT/$O/pmc/p4.log 2>&1; echo rc=$?
cd $GRAFT_REPO_ROOT
for p in p1 p2 p3 p4; do f=$(find $O/pmc/$p -name "*counter_collection.csv" | head -1); echo "== $p $f"; [ -n "$f" ] && python - "$f" <<'PY'
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
find $O/pmc -name "*.csv" -size +200k -delete; find $O/pmc -name "*.log" -size +64k -delete

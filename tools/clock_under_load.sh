#!/bin/bash
# Shader clock and package power while ONE op of tools/run_op.py runs in a loop (rocm-smi sampled every 0.4 s):
#   tools/clock_under_load.sh <op> [seconds]        e.g. gemmbf6, gemmi8, kron64fq, kron128x224, hadq14336
#   FQHIP_OVERLAY=variants/ov_duo3.so tools/clock_under_load.sh kron128x224      an ablation build: W x us per launch = joules per launch
OP=$1; SEC=${2:-6}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python - <<PY &
import sys, time, subprocess
sys.argv = ["run_op.py", "$OP", "1"]
import runpy, torch
ns = runpy.run_path("tools/run_op.py")
fn = ns["fn"]
t0 = time.time(); n = 0
while time.time() - t0 < $SEC:
    for i in range(50): fn(i)
    torch.cuda.synchronize(); n += 50
print(f"$OP: {n} launches in {time.time() - t0:.1f} s = {(time.time() - t0) / n * 1e6:.1f} us per launch (wall, incl. python)")
PY
PID=$!
sleep 2.5   # import + set-up
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Socket Graphics Package Power" | sed "s/^/[$OP] /"
  sleep 0.4
done
wait $PID

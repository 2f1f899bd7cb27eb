#!/usr/bin/env python3
"""What the 20-step wall-clock bracket of bench.py costs beyond the launches themselves: the same 20 launches of the headline kernel between
synchronize() calls, bracket variants side by side (20 repetitions each, medians): (a) bench.py's bracket (two event records inside), (b) no events
inside, (c) no events + a spin on stream.query() in front of the closing synchronize()."""
import os, sys, time, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench

dev = torch.device("cuda:0")
from flatquant_amd import sharding
bc = bench.TimedBroadcast(sharding, dev)
wl = bench.WORKLOADS["C2"](dev, 0, 1, sharding, bc)
step, stream = wl.step, torch.cuda.current_stream(dev)
t = time.perf_counter()
while time.perf_counter() - t < 0.3:
    for i in range(64): step(i)
    torch.cuda.synchronize()

def bracket(kind, K=20):
    for i in range(5): step(i)
    torch.cuda.synchronize(); torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    if kind == "a": ev0.record(stream)
    for i in range(K): step(i)
    if kind == "a": ev1.record(stream)
    if kind == "c":
        while not stream.query(): pass
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e6, (ev0.elapsed_time(ev1) / K * 1e3 if kind == "a" else float("nan"))

res = {k: [] for k in "abc"}
for rep in range(20):
    for k in "abc":
        res[k].append(bracket(k))
for k in "abc":
    w = sorted(x[0] for x in res[k]); e = [x[1] for x in res[k]]
    print(f"bracket {k}: wall per step median {statistics.median(w):.2f} us  (min {w[0]:.2f}, p80 {w[16]:.2f})" + (f"   events {statistics.median(e):.2f} us" if k == "a" else ""), flush=True)
for K in (20, 100, 1000):
    w = sorted(bracket("b", K)[0] for _ in range(7))
    print(f"K = {K}: wall per step median {w[3]:.2f} us", flush=True)

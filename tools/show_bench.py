#!/usr/bin/env python3
"""Print the one-line JSON records of bench.py as a small table: tools/show_bench.py a.json b.json ..."""
import json
import sys

for path in sys.argv[1:]:
    try:
        d = json.load(open(path))
    except Exception as e:  # noqa: BLE001
        print(path, "no json:", e)
        continue
    r = d["roofline"]
    print(f"{path}: {d['value']:.3e} Melem/s, {d['ms_per_step'] * 1e3:.1f} us/step, n_gpus {d['n_gpus']} | dominant: {r['kernel']} "
          f"{r['launch_us']:.1f} us frac {r['frac']:.3f} | step frac {r.get('step', {}).get('frac')}")
    for k in r.get("kernels", []):
        print(f"     {k['kernel']:58s} {k['launch_us']:8.1f} us  frac {k['frac']:.3f}")

#!/usr/bin/env python3
"""Per-kernel table of one bench.py line (stdin): launches x average us, share of the GPU time."""
import json
import sys

d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d["value"]), "graphs/s", round(d["ms_per_step"], 3), "ms/step", d["config"]["workload"][:40])
for n, x in (d["roofline"].get("kernels") or {}).items():
    print(f"{n:28s} {x['launches']:5d} x {x['avg_us']:8.1f} us   {100 * x['share']:5.1f} %")

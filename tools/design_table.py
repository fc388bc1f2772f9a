#!/usr/bin/env python3
"""Rows of DESIGN.md section 8 from profiles/<prefix>_bench_*.json:  python tools/design_table.py [r05_final]"""
import json
import sys

PRE = sys.argv[1] if len(sys.argv) > 1 else "r05_final"
ROWS = [("zinc500k_n64", "`zinc500k_n64` (2, headline: B = 128, N = 64, De = 64, Ly = 10, fp32)"),
        ("driver_style", "the same, `--steps 20 --warmup 10` (the driver's shape)"),
        ("zinc500k_n64_full", "`zinc500k_n64_full` (2 with every node real)"),
        ("zinc100k_n37", "`zinc100k_n37` (1: N = 37, Dh = De = 48, Ly = 4)"),
        ("cifar10_n150", "`cifar10_n150` (3 as specified: bf16 edge tensors)"),
        ("cifar10_n150_fp32", "`cifar10_n150_fp32`"),
        ("pattern500k_n120_b128", "`pattern500k_n120_b128` (4 shapes at B = 128, Ly = 16)"),
        ("pattern500k_n120", "`pattern500k_n120` (4 as specified: B = 16 per GPU)"),
        ("pattern500k_bmax", "`pattern500k_bmax` (4, padded to the per-batch max: N = 182, B = 16)"),
        ("pattern500k_bmax_b128", "`pattern500k_bmax_b128` (N = 188)"),
        ("pattern500k_n188", "`pattern500k_n188` (fixed N = 188, B = 16)"),
        ("pattern500k_n188_b128", "`pattern500k_n188_b128`"),
        ("synthetic_n512", "`synthetic_n512` (5, core-op scope, B = 8; `bound = mfma`)"),
        ("synthetic_n512_b32", "`synthetic_n512_b32`"),
        ("synthetic_n512_block", "`synthetic_n512_block` (5, block scope, FUSED pair operator)"),
        ("synthetic_n512_block_b32", "`synthetic_n512_block_b32` (the same at B = 32)"),
        ("scope_layers", "`--scope layers` (ZINC, attention block + node / edge FFN per layer)"),
        ("scope_model", "`--scope model` (whole ZINC model)")]


def load(n):
    for ln in open(f"profiles/{PRE}_bench_{n}.json"):
        if ln.startswith("{"):
            return json.loads(ln)


import os
for n, label in ROWS:
    if not os.path.exists(f"profiles/{PRE}_bench_{n}.json"):
        continue   # (a workload this profile run did not include)
    d = load(n); r = d["roofline"]; k = r.get("kernels") or {}
    sm = d["config"].get("step_mode"); sm = sm.get("chosen") if isinstance(sm, dict) else (sm or "—")
    nk = 4 if ("synthetic" in n or "scope" in n) else 2
    ks = ", ".join(f"`{a}` {b['avg_us']:.1f} µs" for a, b in list(k.items())[:nk])
    med = d.get("median_ms_per_step") or 0
    extra = ""
    if n == "zinc500k_n64":
        extra = (f" of 8 TB/s ({r.get('frac_of_achievable', 0):.2f} of 6.29; from the launches sampled INSIDE the timed region: {r['avg_launch_us']:.1f} µs);"
                 f" issue {r['issue']['frac']:.2f}, matrix pipe {r['mfma_busy']:.2f}; PMC traffic {r['traffic'] / 1e6:.0f} / {r['algorithmic_bytes_per_launch'] / 1e6:.0f} MB")
    if r.get("unit") == "TFLOP/s":
        extra = f" of 157.3 TF (`{r['kernel']}`)"
    print(f"| {label} | **{d['value'] / 1e3:.1f} k** ({sm}) | {d['ms_per_step']:.3f} / {med:.3f} | {ks} | {r['frac']:.3f}{extra} |")

#!/usr/bin/env python3
"""Randomised parity sweep of the De = 8 pair kernels (egt_narrow.hip) at stack scope: random N (ragged tiles, shared
tiles of the balanced ranges, single-row groups), batch, feature variant and edge dtype against the fp64 oracle
(tests/test_narrow_gpu.py::check_de8_stack).  The kernel-selection switches are read once per process, so every switch
setting runs in its own child process:   python tools/sweep_de8.py [cases-per-setting]      (EGT_SWEEP_SEED=<n>)"""
import os
import random
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SETTINGS = [{}, {"EGT_BWD_TL": "16", "EGT_NRW_FWD_WAVES": "4", "EGT_NRW_BWD_WAVES": "4"}, {"EGT_BWD_TL": "11", "EGT_NRW_BWD_WAVES": "8"}, {"EGT_BWD_TL": "5", "EGT_NRW_FWD_WAVES": "8", "EGT_NRW_FWD_HALF": "0"}, {"EGT_NRW_FWD_WAVES": "8", "EGT_NRW_FWD_HALF": "1"}]


def child(ncase, seed):
    import torch
    sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, REPO)
    from test_narrow_gpu import check_de8_stack
    rng = random.Random(seed)
    gpu = torch.device("cuda", 0)
    bad = 0
    for i in range(ncase):
        variant = rng.choice(["plain", "plain", "plain", "ungated", "noclip", "bias"])
        N = rng.choice([rng.randint(2, 40), rng.randint(41, 130), rng.randint(131, 200)])
        B = rng.choice([1, 2, 3]) if N > 100 else rng.choice([1, 2, 5, 9])
        bf16, train = rng.random() < 0.5, rng.random() < 0.6
        try:
            check_de8_stack(variant, N, 64, bf16, train, gpu, B=B, Ly=rng.choice([1, 2, 3, 4]))
        except AssertionError as e:  # noqa: PERF203
            bad += 1
            print(f"FAIL variant={variant} N={N} B={B} bf16={bf16} train={train}: {str(e)[:300]}", flush=True)
    print(f"de8 sweep child: {ncase} cases, {bad} failures", flush=True)
    return bad


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        sys.exit(1 if child(int(sys.argv[2]), int(sys.argv[3])) else 0)
    ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    seed = int(os.environ.get("EGT_SWEEP_SEED", "0"))
    tot = 0
    for k, st in enumerate(SETTINGS):
        r = subprocess.run([sys.executable, __file__, "--child", str(ncase), str(seed * 100 + k)], env=dict(os.environ, **st),
                           cwd=REPO, capture_output=True, text=True)
        out = [ln for ln in r.stdout.splitlines() if ln.startswith(("FAIL", "de8 sweep"))]
        print(f"setting {st or 'default'}: " + (" | ".join(out) if out else (r.stdout + r.stderr)[-400:]), flush=True)
        tot += r.returncode != 0
    print(f"sweep: De = 8 stacks, {len(SETTINGS)} switch settings x {ncase} random geometries, settings with failures: {tot}")
    sys.exit(1 if tot else 0)


if __name__ == "__main__":
    main()

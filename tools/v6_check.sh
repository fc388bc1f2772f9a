#!/bin/bash
# GPU-box check of the tile-pair backward (k_block_bwd_v6): parity suites that reach it, then A/B bench vs v5.
export EGT_BWD_V6=1
timeout 900 python -m pytest tests/test_block_gpu.py tests/test_fullsize_gpu.py tests/test_graph_gpu.py -m gpu -x -q 2>&1 | tail -8
for v in 0 1; do
  EGT_BWD_V6=$v timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); k=d['roofline']['kernels']
        print('V6=$v', round(d['value']), 'graphs/s', {n:round(x['avg_us'],1) for n,x in list(k.items())[:3]})
"
done

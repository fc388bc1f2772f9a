#!/bin/bash
# round 5: non-temporal input loads -- layers / model scope with the FFN hint on (in-tree) and off (ffnplain), prev = before any hint
out=gpurun_out/r05_nt; mkdir -p $out
for v in default ffnplain prev default ffnplain prev; do
  if [ "$v" = default ]; then unset EGT_AMD_LIB; else export EGT_AMD_LIB=$PWD/egt_amd/lib/var/libegt_$v.so; fi
  for sc in layers model; do
    timeout 300 python bench.py --scope $sc --no-cpu-baseline --no-graph-leg --steps 20 2> $out/err_${v}_$sc.log | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); r=d['roofline'] or {}; k=r.get('kernels') or {}
        print('$v $sc', round(d['value']), 'graphs/s', round(d['ms_per_step'],3), 'ms frac', round(r.get('frac') or 0,3), {n:round(x['avg_us'],1) for n,x in list(k.items())[:4]})
" | tee -a $out/ab.txt
  done
done
unset EGT_AMD_LIB
timeout 1500 python -m pytest tests/test_ffn_gpu.py tests/test_block_gpu.py tests/test_fullsize_gpu.py tests/test_graph_gpu.py tests/test_model.py -x -q -m gpu 2>&1 | tail -3

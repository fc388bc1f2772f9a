#!/bin/bash
# round 5: A/B on one box -- (a) headline: in-tree vs the two-accumulator P1/P2 chains (twoacc) vs the previous library (prev);
# (b) layers / model scope: FFN backward without spills + two-accumulator chains (in-tree) vs prev
out=gpurun_out/r05_ffn; mkdir -p $out
tools/ab.sh "--no-graph-leg --graph off --steps 50" default twoacc prev default twoacc 2>&1 | tee $out/ab_headline.txt
for v in default prev default prev; do
  if [ "$v" = default ]; then unset EGT_AMD_LIB; else export EGT_AMD_LIB=$PWD/egt_amd/lib/var/libegt_$v.so; fi
  for sc in layers model; do
    timeout 300 python bench.py --scope $sc --no-cpu-baseline --no-graph-leg --steps 20 2> $out/err_${v}_$sc.log | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); r=d['roofline'] or {}; k=r.get('kernels') or {}
        print('$v $sc', round(d['value']), 'graphs/s', round(d['ms_per_step'],3), 'ms frac', round(r.get('frac') or 0,3), {n:round(x['avg_us'],1) for n,x in list(k.items())[:4]})
" | tee -a $out/ab_ffn.txt
  done
done
unset EGT_AMD_LIB
timeout 900 python -m pytest tests/test_ffn_gpu.py tests/test_model.py -x -q -m gpu 2>&1 | tail -3

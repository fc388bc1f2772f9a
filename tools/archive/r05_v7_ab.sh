#!/bin/bash
# round 5: k_block_bwd_v7 -- parity on the box, then the headline A/B (EGT_BWD_V7=0 = k_block_bwd_v5)
out=gpurun_out/r05_v7; mkdir -p $out
timeout 900 python -m pytest tests/test_bwd_v7_gpu.py -x -q -m gpu > $out/pytest_v7.log 2>&1; echo "pytest v7 rc=$?" | tee -a $out/summary.txt
tail -5 $out/pytest_v7.log
timeout 900 python -m pytest tests/test_fullsize_gpu.py tests/test_graph_gpu.py -x -q -m gpu > $out/pytest_full.log 2>&1; echo "pytest fullsize rc=$?" | tee -a $out/summary.txt
tail -3 $out/pytest_full.log
for v in 0 1 default; do
  if [ $v = default ]; then unset EGT_BWD_V7; else export EGT_BWD_V7=$v; fi
  for rep in 1 2; do
  timeout 600 python bench.py --no-cpu-baseline --no-graph-leg --graph off --steps 50 2> $out/bench_v7_${v}_$rep.err | tee $out/bench_v7_${v}_$rep.json | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); r=d['roofline'] or {}; k=r.get('kernels') or {}
        print('V7=$v rep$rep', round(d['value']), 'graphs/s', round(d['ms_per_step'],3), 'ms', {n:round(x['avg_us'],1) for n,x in list(k.items())[:4]})
" | tee -a $out/summary.txt
  done
done

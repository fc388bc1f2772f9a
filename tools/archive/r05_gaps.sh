#!/bin/bash
# round 5: how much of a headline step is idle time BETWEEN kernels (eager launches and hipGraph replay)
export TMPDIR=/tmp
out=gpurun_out/r05_gaps; mkdir -p $out
for g in off on; do
  timeout 600 rocprofv3 --kernel-trace -d $out/kt_$g -o r -- python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-prof --no-graph-leg --graph $g > $out/bench_$g.json 2> $out/err_$g.log
  echo "== graph $g" | tee -a $out/gaps.txt
  python -c "
import json
for ln in open('$out/bench_$g.json'):
    if ln.startswith('{'):
        d=json.loads(ln); print('bench', round(d['value']), 'graphs/s', d['ms_per_step'], 'ms/step')" | tee -a $out/gaps.txt
  python tools/gap_stat.py $out/kt_$g 0.5 | tee -a $out/gaps.txt
done
rm -rf $out/kt_*

#!/bin/bash
# Measurement table of SURVEY.md §8(d): every config's shapes at core-op, single-block and
# stack scope (fp32).  Run through gpurun; prints one JSON line per row.
for w in zinc500k_n64 zinc100k_n37 cifar10_n150_fp32 cifar10_n150 cifar10_n150_pad160 pattern500k_n120 pattern500k_n120_b128 pattern500k_n120_pad128_b128; do
  for ly in 1 0; do
    timeout 300 python bench.py --workload $w --layers $ly --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); c=d['config']; r=d.get('roofline') or {}
print(json.dumps(dict(workload='$w', Ly=c.get('Ly'), path=c.get('path'), graphs_per_s=round(d['value'],1), ms_per_step=round(d['ms_per_step'],4),
      kernels={k:round(v['avg_us'],1) for k,v in (r.get('kernels') or {}).items()})))"
  done
done
for c in cfg2 cfg4 cfg5; do timeout 200 python tools/bench_core.py $c 2>/dev/null | tail -1; done

#!/bin/bash
OUT=gpurun_out/r06_pair2; mkdir -p $OUT
timeout 300 python bench.py --workload synthetic_n512_block --no-cpu-baseline > $OUT/bench_block.json 2> $OUT/bench_block_err.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06_pair2/bench_block.json').read().strip().splitlines()[-1])
print(round(d['value']), d['ms_per_step'])
r=d['roofline']
for k,v in r['kernels'].items(): print(k, v)
print({k:v for k,v in r.items() if k!='kernels'})
PY

#!/bin/bash
# parity of the pair kernels + two block-scope lines (current build)
OUT=gpurun_out/r06_quick; mkdir -p $OUT
timeout 900 python -m pytest tests/test_pair_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q -k "pair or n512_block" > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
for rep in 1 2; do
  timeout 300 python bench.py --workload synthetic_n512_block --no-cpu-baseline > $OUT/b_$rep.json 2>> $OUT/err.log
  python - $OUT/b_$rep.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=(d.get('roofline') or {}).get('kernels') or {}
print(round(d['value']), round(d['ms_per_step'],4), {n:round(v['avg_us'],1) for n,v in list(k.items())[:3]})
PY
done

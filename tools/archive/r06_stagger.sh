#!/bin/bash
OUT=gpurun_out/r06_stagger; mkdir -p $OUT
python -c "from egt_amd import build as B; B.build()" > $OUT/build.log 2>&1
timeout 900 python -m pytest tests/test_pair_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q -k "pair or n512_block" > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
line() { python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=(d.get('roofline') or {}).get('kernels') or {}
print(sys.argv[2], round(d['value']), round(d['ms_per_step'],4), {n:round(v['avg_us'],1) for n,v in list(k.items())[:3]})
PY
}
for rep in 1 2; do
  timeout 300 python bench.py --workload synthetic_n512_block --no-cpu-baseline > $OUT/b_stag_$rep.json 2>> $OUT/err.log; line $OUT/b_stag_$rep.json stagger
done
EGT_ATTN_FLAGS="-DPAIR_STAGGER=0" python -c "from egt_amd import build as B; B.build()" >> $OUT/build.log 2>&1
for rep in 1 2; do
  EGT_ATTN_FLAGS="-DPAIR_STAGGER=0" timeout 300 python bench.py --workload synthetic_n512_block --no-cpu-baseline > $OUT/b_lock_$rep.json 2>> $OUT/err.log; line $OUT/b_lock_$rep.json lockstep
done

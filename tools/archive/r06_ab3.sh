#!/bin/bash
OUT=gpurun_out/r06_ab3; mkdir -p $OUT
line() { python - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=(d.get('roofline') or {}).get('kernels') or {}
print(sys.argv[2], round(d['value']), round(d['ms_per_step'],4), {n:round(v['avg_us'],1) for n,v in list(k.items())[:3]})
PY
}
for f in "-DPAIR_BWD_NT=0" "-DPAIR_BWD_NT=1" "-DPAIR_BWD_NT=5" "-DPAIR_BWD_NT=4"; do
  EGT_ATTN_FLAGS="$f" python -c "from egt_amd import build as B; B.build()" > $OUT/build.log 2>&1
  for rep in 1 2; do
    EGT_ATTN_FLAGS="$f" timeout 300 python bench.py --workload synthetic_n512_block --no-cpu-baseline > $OUT/b.json 2>> $OUT/err.log; line $OUT/b.json "$f"
  done
done

#!/bin/bash
# same-box A/B of k_pair_bwd build variants (egt_amd/lib/var/libegt_<name>.so): bwd / fwd us per launch, graphs/s
OUT=gpurun_out/r06_abl2; mkdir -p $OUT
V=$PWD/egt_amd/lib/var
for rep in ${REPS:-1 2}; do
  for v in "$@"; do
    L=$V/libegt_$v.so; [ $v = new ] && L=""
    EGT_AMD_LIB=$L timeout 300 python bench.py --workload ${WL:-synthetic_n512_block} --no-cpu-baseline > $OUT/b_${v}_$rep.json 2>> $OUT/err.log
    python - $OUT/b_${v}_$rep.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=(d.get('roofline') or {}).get('kernels') or {}
print(sys.argv[1].split('/')[-1], round(d['value']), round(d['ms_per_step'],4), {n:round(v['avg_us'],1) for n,v in list(k.items())[:3]})
PY
  done
done

#!/bin/bash
# k_pair_bwd: LDS layouts without bank conflicts (row skew of the edge waves, scratch [t][pair][16], operand rows swapped) and the phased POST:
# parity of both builds, same-box A/B against the previous build (egt_amd/lib/var/libegt_prev.so), phase stamps
OUT=gpurun_out/r06_post4; mkdir -p $OUT
V=$PWD/egt_amd/lib/var
timeout 900 python -m pytest tests/test_pair_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q -k "pair or n512_block" > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
EGT_AMD_LIB=$V/libegt_post0.so timeout 900 python -m pytest tests/test_pair_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q -k "pair or n512_block" > $OUT/pytest0.log 2>&1; echo "rc=$?" >> $OUT/pytest0.log
tail -3 $OUT/pytest0.log
line() { python - $1 <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=(d.get('roofline') or {}).get('kernels') or {}
print(sys.argv[1].split('/')[-1], round(d['value']), round(d['ms_per_step'],4), {n:round(v['avg_us'],1) for n,v in list(k.items())[:3]})
PY
}
for rep in 1 2; do
  for v in new post0 prev; do
    L=""; [ $v != new ] && L=$V/libegt_$v.so
    EGT_AMD_LIB=$L timeout 300 python bench.py --workload synthetic_n512_block --no-cpu-baseline > $OUT/b_${v}_$rep.json 2>> $OUT/err.log
    line $OUT/b_${v}_$rep.json
  done
done
for v in 4 0; do
EGT_AMD_LIB=$V/libegt_stamps$v.so timeout 300 python tools/pair_stamps.py > $OUT/stamps$v.txt 2>&1
echo "stamps POST4=$v"; grep -A 30 "^k_pair_bwd" $OUT/stamps$v.txt
done

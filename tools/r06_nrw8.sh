#!/bin/bash
# k_narrow_bwd with eight waves per workgroup (launches of <= one workgroup per CU): parity, then config 4 as specified (B = 16) with 4 / 8 waves on one box
OUT=gpurun_out/r06_nrw8; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_narrow_gpu.py -m gpu -x -q -k "waves_per_workgroup or half_row or test_random_de8" > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
for rep in 1 2; do for w in 0 1; do
  EGT_NRW_FWD_HALF=$w timeout 300 python bench.py --workload pattern500k_n120 --no-cpu-baseline > $OUT/b16_w${w}_$rep.json 2>> $OUT/err.log
  python - $OUT/b16_w${w}_$rep.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=(d.get('roofline') or {}).get('kernels') or {}
print(sys.argv[1].split('/')[-1], round(d['value']), round(d['ms_per_step'],4), d['config'].get('step_mode',{}).get('chosen'), {n:round(v['avg_us'],1) for n,v in list(k.items())[:4]})
PY
done; done
for wl in zinc100k_n37 pattern500k_n120_b128 cifar10_n150; do
  timeout 300 python bench.py --workload $wl --no-cpu-baseline > $OUT/b_$wl.json 2>> $OUT/err.log
  python - $OUT/b_$wl.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=(d.get('roofline') or {}).get('kernels') or {}
print(sys.argv[1].split('/')[-1], round(d['value']), round(d['ms_per_step'],4), {n:round(v['avg_us'],1) for n,v in list(k.items())[:2]})
PY
done

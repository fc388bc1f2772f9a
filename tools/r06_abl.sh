#!/bin/bash
# round 6: timing ablations of k_pair_fwd (results wrong by design; EGT_ATTN_FLAGS=-DPAIR_ABL=<bits>): which role bounds the launch
OUT=gpurun_out/r06_abl; mkdir -p $OUT
for abl in 0 2 16; do
  EGT_ATTN_FLAGS="-DPAIR_ABL=$abl" python -c "from egt_amd import build as B; B.build()" > $OUT/build.log 2>&1
  EGT_ATTN_FLAGS="-DPAIR_ABL=$abl" timeout 300 python bench.py --workload synthetic_n512_block_nomask --no-cpu-baseline --steps 20 > $OUT/b_$abl.json 2>> $OUT/err.log
  python - $OUT/b_$abl.json $abl <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print('PAIR_ABL', sys.argv[2], 'fwd', round(k['k_pair_fwd']['avg_us'],1), 'bwd', round(k['k_pair_bwd']['avg_us'],1))
PY
done

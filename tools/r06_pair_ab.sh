#!/bin/bash
# round 6: build-flag A/Bs of the pair kernels on one box (EGT_ATTN_FLAGS variants), block-scope line each
OUT=gpurun_out/r06_pair_ab; mkdir -p $OUT
run() {
  tag=$1; shift
  EGT_ATTN_FLAGS="$*" python -c "from egt_amd import build as B; B.build()" > $OUT/build_$tag.log 2>&1
  for rep in 1 2; do
    EGT_ATTN_FLAGS="$*" timeout 300 python bench.py --workload synthetic_n512_block --no-cpu-baseline > $OUT/bench_${tag}_$rep.json 2>> $OUT/err.log
    python - "$OUT/bench_${tag}_$rep.json" "$tag" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d['roofline']['kernels']
print(sys.argv[2], round(d['value']), round(d['ms_per_step'],4), 'fwd', round(k['k_pair_fwd']['avg_us'],1), 'bwd', round(k['k_pair_bwd']['avg_us'],1))
PY
  done
}
run base -DPAIR_EDGE_PRIO=2
run nt -DPAIR_NT_E=1
run prio3 -DPAIR_EDGE_PRIO=3
run prio1 -DPAIR_EDGE_PRIO=1

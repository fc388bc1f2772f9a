#!/bin/bash
# A/B of the De = 8 VALU pair kernels against r4 / v4r on the narrow workloads (run through gpurun).
#   bash tools/narrow_ab.sh [quick]   -> gpurun_out/nrw/bench.log (+ pytest.log unless quick)
mkdir -p gpurun_out/nrw
: > gpurun_out/nrw/bench.log
if [ "${1:-}" != "quick" ]; then
  timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/nrw/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/nrw/pytest.log
  tail -3 gpurun_out/nrw/pytest.log
fi
WL=${NRW_WORKLOADS:-"cifar10_n150_fp32 cifar10_n150 pattern500k_n120 pattern500k_n120_b128"}
for w in $WL; do
  for nn in ${NRW_ARMS:-0 1}; do
    EGT_NO_NARROW=$nn EGT_NO_NARROW_BWD=${NRW_NOBWD:-0} timeout 300 python bench.py --workload $w --no-cpu-baseline --steps 30 --warmup 5 2>>gpurun_out/nrw/err.log | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); k = d['roofline']['kernels']
        print('$w no_narrow=$nn no_bwd=${NRW_NOBWD:-0}', round(d['value']), 'graphs/s', round(d['ms_per_step'], 3), 'ms |', ' '.join(f'{n}={v[\"avg_us\"]:.1f}x{v[\"launches\"]}' for n, v in k.items()))
" >> gpurun_out/nrw/bench.log
  done
done
cat gpurun_out/nrw/bench.log

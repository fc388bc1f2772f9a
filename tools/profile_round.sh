#!/bin/bash
# Round-end measurement on the GPU box (run through gpurun):
#   bash tools/profile_round.sh <tag>
# Writes everything under gpurun_out/<tag>/ ; tools/prof_summary.py turns it into profiles/<tag>_*.
# Counter passes are separate from each other and carry --kernel-trace only (no sys/hip/hsa traces).
set -u
TAG=${1:-final}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-prof"
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench_err.log
timeout 300 python tools/bench_core.py cfg5 > $OUT/core_cfg5.json 2>> $OUT/bench_err.log
timeout 300 python tools/bench_core.py cfg2 > $OUT/core_cfg2.json 2>> $OUT/bench_err.log
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/kt -o r -- $BENCH > $OUT/bench_under_rocprof.json 2>> $OUT/bench_err.log
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o r -- $BENCH > /dev/null 2>> $OUT/bench_err.log
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o r -- $BENCH > /dev/null 2>> $OUT/bench_err.log
timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/pmc_sq -o r -- $BENCH > /dev/null 2>> $OUT/bench_err.log
# HBM traffic of the De = 8 workloads (BASELINE configs 3 and 4 at B = 128) for roofline.traffic of their bench lines
for WL in cifar10_n150 pattern500k_n120_b128; do
  B2="python bench.py --workload $WL --steps 6 --warmup 2 --no-cpu-baseline --no-prof --no-graph-leg"
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch_$WL -o r -- $B2 > /dev/null 2>> $OUT/bench_err.log
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write_$WL -o r -- $B2 > /dev/null 2>> $OUT/bench_err.log
  timeout 300 python bench.py --workload $WL --no-cpu-baseline > $OUT/bench_$WL.json 2>> $OUT/bench_err.log
done
find $OUT -name "*.db" | head
tail -3 $OUT/pytest_gpu.log

#!/bin/bash
# Round-4 measurement on the GPU box (run through gpurun):  bash tools/profile_r04.sh <tag> [quick]
# Writes everything under gpurun_out/<tag>/ ; tools/prof_r04_summary.py turns it into profiles/r04_<tag>_*.
# Counter passes are separate from each other and carry --kernel-trace only (no sys / hip / hsa traces).
set -u
TAG=${1:-final}
QUICK=${2:-}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
rocprofv3 -L > $OUT/counters_list.txt 2>&1 || true
if [ -z "$QUICK" ]; then
  timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
fi
for WL in zinc500k_n64 synthetic_n512; do
  timeout 400 python bench.py --workload $WL > $OUT/bench_$WL.json 2> $OUT/bench_${WL}_err.log
  B="python bench.py --workload $WL --steps 10 --warmup 3 --no-cpu-baseline --no-prof --no-graph-leg --graph off"   # counters per eager launch
  timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/kt_$WL -o r -- $B > $OUT/bench_under_rocprof_$WL.json 2>> $OUT/bench_${WL}_err.log
  timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch_$WL -o r -- $B > /dev/null 2>> $OUT/bench_${WL}_err.log
  timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write_$WL -o r -- $B > /dev/null 2>> $OUT/bench_${WL}_err.log
  # matrix-pipe busy, wave cycles and stalls (8 SQ slots) + the GPU-active cycle count (GRBM: its own block)
  timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $OUT/pmc_sq_$WL -o r -- $B > /dev/null 2>> $OUT/bench_${WL}_err.log
  # instruction counts by class: the issue roof (MFMA cycles + VALU cycles per SIMD)
  timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES -d $OUT/pmc_inst_$WL -o r -- $B > /dev/null 2>> $OUT/bench_${WL}_err.log
done
if [ -z "$QUICK" ]; then
  for WL in cifar10_n150 pattern500k_n120_b128 zinc100k_n37 pattern500k_n120 synthetic_n512_b32 synthetic_n512_block; do
    timeout 300 python bench.py --workload $WL --no-cpu-baseline > $OUT/bench_$WL.json 2>> $OUT/bench_err.log
  done
  timeout 300 python tools/bench_block_cfg5.py > $OUT/block_cfg5.json 2>> $OUT/bench_err.log
fi
find $OUT -name "*.db" | head -20

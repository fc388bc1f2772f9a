#!/bin/bash
# De = 8 kernels on BASELINE config 3 as specified (bf16, N = 150): kernel trace + PMC passes (run through gpurun)
#   bash tools/profile_narrow.sh <tag>  -> gpurun_out/<tag>/{kt_cifar,pmc_sq_cifar,pmc_fetch_cifar,pmc_write_cifar}
set -u
TAG=${1:-narrow}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
BENCH="python bench.py --workload cifar10_n150 --steps 10 --warmup 3 --no-cpu-baseline --no-prof"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/kt_cifar -o r -- $BENCH > $OUT/bench_cifar.json 2> $OUT/err.log
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE -d $OUT/pmc_sq_cifar -o r -- $BENCH > /dev/null 2>> $OUT/err.log
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch_cifar -o r -- $BENCH > /dev/null 2>> $OUT/err.log
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write_cifar -o r -- $BENCH > /dev/null 2>> $OUT/err.log
find $OUT -name "*.db" | head

#!/bin/bash
# PMC evidence for the MFMA-bound kernels (channel FFN, MFMA inner op): gpurun_out/extra/*.db
set -u
export TMPDIR=/tmp
OUT=gpurun_out/extra
mkdir -p $OUT
C1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS"
C2="SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/ffn_kt -o r -- python tools/bench_ffn.py > $OUT/ffn.json 2> $OUT/err.log
timeout 300 rocprofv3 --kernel-trace --pmc $C1 -d $OUT/ffn_a -o r -- python tools/bench_ffn.py > /dev/null 2>> $OUT/err.log
timeout 300 rocprofv3 --kernel-trace --pmc $C2 -d $OUT/ffn_b -o r -- python tools/bench_ffn.py > /dev/null 2>> $OUT/err.log
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/core_kt -o r -- python tools/bench_core.py cfg5 > $OUT/core.json 2>> $OUT/err.log
timeout 300 rocprofv3 --kernel-trace --pmc $C1 -d $OUT/core_a -o r -- python tools/bench_core.py cfg5 > /dev/null 2>> $OUT/err.log
timeout 300 rocprofv3 --kernel-trace --pmc $C2 -d $OUT/core_b -o r -- python tools/bench_core.py cfg5 > /dev/null 2>> $OUT/err.log
find $OUT -name "*.db" | wc -l

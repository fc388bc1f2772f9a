#!/bin/bash
# Round-6 measurement on the GPU box (run through gpurun):  bash tools/profile_r06.sh <tag> [quick]
# Everything lands under gpurun_out/<tag>/; the rocprofv3 databases are summarised ON THE BOX (tools/prof_r04_summary.py writes
# profiles/r06_<tag>_* into the box's copy of the tree), the summaries are copied to gpurun_out/<tag>/profiles/ and the databases
# deleted (gpurun merges at most 64 MiB back).  Counter passes are separate from each other and carry --kernel-trace only.
set -u
TAG=${1:-final}
QUICK=${2:-}
OUT=gpurun_out/$TAG
mkdir -p $OUT/profiles
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
if [ -z "$QUICK" ]; then
  timeout 2700 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
fi
FULLPMC="zinc500k_n64 synthetic_n512 synthetic_n512_block synthetic_n512_block_b32"
HBMPMC="cifar10_n150 pattern500k_n120 pattern500k_n120_b128 zinc100k_n37"   # fresh HBM-byte counters for the lines whose entries dated from round 3
export PROF_WORKLOADS="$FULLPMC $HBMPMC"
for WL in $FULLPMC $HBMPMC; do
  : > $OUT/bench_${WL}_err.log
  B="python bench.py --workload $WL --steps 10 --warmup 3 --no-cpu-baseline --no-prof --no-graph-leg --graph off"   # counters per eager launch
  timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/kt_$WL -o r -- $B > $OUT/bench_under_rocprof_$WL.json 2>> $OUT/bench_${WL}_err.log
  timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch_$WL -o r -- $B > /dev/null 2>> $OUT/bench_${WL}_err.log
  timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write_$WL -o r -- $B > /dev/null 2>> $OUT/bench_${WL}_err.log
  case " $FULLPMC " in *" $WL "*)
    timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $OUT/pmc_sq_$WL -o r -- $B > /dev/null 2>> $OUT/bench_${WL}_err.log
    timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES -d $OUT/pmc_inst_$WL -o r -- $B > /dev/null 2>> $OUT/bench_${WL}_err.log ;;
  esac
done
# counters first, bench lines second: a bench line cites profiles/pmc_*.json (traffic, matrix-pipe busy), which the summary tool
# refreshes from THIS run's counter passes
python tools/prof_r04_summary.py $OUT r06_$TAG > $OUT/summary_pre.txt 2>&1
for WL in $FULLPMC $HBMPMC; do
  timeout 400 python bench.py --workload $WL $( [ $WL = zinc500k_n64 ] || echo --no-cpu-baseline ) > $OUT/bench_$WL.json 2>> $OUT/bench_${WL}_err.log
done
# the driver's own invocation shape, and the step modes side by side
timeout 300 python bench.py --steps 20 --warmup 10 > $OUT/bench_driver_style.json 2>> $OUT/bench_err.log
timeout 300 python bench.py --graph on --no-cpu-baseline --no-graph-leg > $OUT/bench_graph_on.json 2>> $OUT/bench_err.log
timeout 300 python bench.py --graph off --no-cpu-baseline --no-graph-leg > $OUT/bench_graph_off.json 2>> $OUT/bench_err.log
if [ -z "$QUICK" ]; then
  for WL in synthetic_n512_b32 zinc500k_n64_full pattern500k_bmax pattern500k_bmax_b128 pattern500k_n188 pattern500k_n188_b128 cifar10_n150_fp32; do
    timeout 300 python bench.py --workload $WL --no-cpu-baseline > $OUT/bench_$WL.json 2>> $OUT/bench_err.log
  done
  for SC in layers model; do
    timeout 300 python bench.py --scope $SC --no-cpu-baseline > $OUT/bench_scope_$SC.json 2>> $OUT/bench_err.log
  done
fi
python tools/prof_r04_summary.py $OUT r06_$TAG > $OUT/summary_tail.txt 2>&1
cp profiles/r06_${TAG}_* profiles/pmc_traffic.json profiles/pmc_mfma.json $OUT/profiles/ 2>/dev/null
rm -rf $OUT/kt_* $OUT/pmc_*
ls $OUT/profiles | head -40

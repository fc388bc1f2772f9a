#!/bin/bash
# round 5: counter passes of the headline backward, k_block_bwd_v5 (default) against k_block_bwd_v7 (EGT_BWD_V7=2), same box.
# Counter passes are separate from each other and carry --kernel-trace only.
set -u
OUT=gpurun_out/r05_v7pmc; mkdir -p $OUT
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-prof --no-graph-leg --graph off"
for v in 0 2; do
  export EGT_BWD_V7=$v
  timeout 300 python bench.py --no-cpu-baseline --no-graph-leg --graph off --steps 50 > $OUT/bench_v7_$v.json 2> $OUT/bench_v7_$v.err
  timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/kt_$v -o r -- $B > /dev/null 2>> $OUT/err.log
  timeout 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d $OUT/pmc_sq_$v -o r -- $B > /dev/null 2>> $OUT/err.log
  timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES -d $OUT/pmc_inst_$v -o r -- $B > /dev/null 2>> $OUT/err.log
  timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch_$v -o r -- $B > /dev/null 2>> $OUT/err.log
  timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write_$v -o r -- $B > /dev/null 2>> $OUT/err.log
done
python - <<'PY'
import glob, sqlite3, os, json
out = {}
for v in ("0", "2"):
    row = {}
    for sub in ("pmc_sq", "pmc_inst", "pmc_fetch", "pmc_write"):
        f = sorted(glob.glob(f"gpurun_out/r05_v7pmc/{sub}_{v}/**/*.db", recursive=True))
        if not f: continue
        db = sqlite3.connect(f[0])
        for kn, cn, c, val, d in db.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection group by kernel_name, counter_name"):
            if "k_block_bwd" in kn:
                row[cn] = val; row.setdefault("dur_us_" + sub, d / 1e3); row["kernel"] = kn[:60]; row["launches_" + sub] = c
    out["v7" if v == "2" else "v5"] = row
json.dump(out, open("gpurun_out/r05_v7pmc/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $OUT/kt_* $OUT/pmc_*     # the rocprofv3 databases stay on the box (gpurun merges at most 64 MiB back)

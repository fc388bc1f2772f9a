#!/bin/bash
# rocprofv3 kernel trace of a hipGraph-replayed step (GPU box, through gpurun):
#   bash tools/profile_graph.sh "<bench args>" <tag>      -> gpurun_out/<tag>.md
set -u
ARGS=${1:-"--scope model --workload pattern500k_n120 --overlap-ffn"}
TAG=${2:-graph_prof}
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
CMD="python bench.py $ARGS --graph on --steps 20 --warmup 3 --no-cpu-baseline --no-prof --no-graph-leg"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt -o r -- $CMD > $OUT/bench.json 2> $OUT/err.log
python - "$OUT" "$CMD" <<'PY' > gpurun_out/$TAG.md
import glob, json, os, sqlite3, sys
out, cmd = sys.argv[1], sys.argv[2]
print(f"# rocprofv3 --kernel-trace --stats -- {cmd}\n")
try:
    d = json.loads([l for l in open(os.path.join(out, "bench.json")) if l.startswith("{")][-1])
    print(f"bench line under the profiler: {d['value']:.1f} graphs/s, {d['ms_per_step']:.3f} ms per step ({d['config']['hipgraph']})\n")
except Exception as e:
    print(f"(bench line unreadable: {e})\n")
f = sorted(glob.glob(os.path.join(out, "kt", "**", "*.db"), recursive=True))
if f:
    db = sqlite3.connect(f[0])
    rows = db.execute("select name, count(*), sum(duration), avg(duration) from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print("| kernel | calls | total_us | avg_us | % |\n|---|---|---|---|---|")
    for n, c, s, a in rows[:24]:
        print(f"| {n[:90]} | {c} | {s / 1e3:.1f} | {a / 1e3:.2f} | {100 * s / tot:.2f} |")
    print(f"\nall kernels: {tot / 1e3:.1f} us over the run")
else:
    print("(no rocprofv3 database)")
PY
cat gpurun_out/$TAG.md | head -40

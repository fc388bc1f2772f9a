#!/usr/bin/env python3
"""Turn the rocprofv3 databases of tools/profile_round.sh into the committed summaries:
    python tools/prof_summary.py gpurun_out/<tag> r01_<tag>
writes profiles/<name>_rocprof_summary.md, profiles/<name>_bench.json and refreshes
profiles/pmc_traffic.json (HBM bytes per launch of the two pair kernels, corrected as
/opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE/WRITE_SIZE are KB, and on gfx950
FETCH_SIZE reports half of a wide coalesced read, so fetch bytes are doubled)."""
import glob
import json
import os
import sqlite3
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def db_of(d):
    f = sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True))
    return sqlite3.connect(f[0]) if f else None


def short(name):
    return name if len(name) <= 100 else name[:100]


def main():
    src, name = sys.argv[1], sys.argv[2]
    out = [f"# {name} — MI355X, measured by tools/profile_round.sh\n"]
    bench = None
    try:
        bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
        out.append("## `python bench.py` (defaults)\n\n```json\n" + json.dumps(bench, indent=1) + "\n```\n")
        json.dump(bench, open(os.path.join(REPO, "profiles", f"{name}_bench.json"), "w"))
    except Exception as e:  # noqa: BLE001
        out.append(f"(bench.json unreadable: {e})\n")
    for core in ("core_cfg2.json", "core_cfg5.json"):
        try:
            line = open(os.path.join(src, core)).read().strip().splitlines()[-1]
            out.append(f"## core-op scope `tools/bench_core.py {core[5:9]}`\n\n```json\n{line}\n```\n")
        except Exception:  # noqa: BLE001
            pass
    db = db_of(os.path.join(src, "kt"))
    if db:
        out.append("## rocprofv3 --kernel-trace --stats -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-prof\n")
        out.append("| kernel | calls | total_us | avg_us | % |\n|---|---|---|---|---|")
        rows = db.execute("select name, count(*), sum(duration), avg(duration) from kernels group by name order by 3 desc").fetchall()
        tot = sum(r[2] for r in rows)
        for n, c, s, a in rows[:16]:
            out.append(f"| {short(n)} | {c} | {s / 1e3:.1f} | {a / 1e3:.2f} | {100 * s / tot:.2f} |")
        out.append("")
    traffic = {}
    out.append("## rocprofv3 --pmc <counters> --kernel-trace (separate passes), per-launch averages\n")
    out.append("| kernel | counter | launches | avg value | avg duration (ns) |\n|---|---|---|---|---|")
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
        db = db_of(os.path.join(src, sub))
        if not db:
            continue
        rows = db.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
                          "where kernel_name like '%k_block_%' or kernel_name like '%k_node_bwd%' or kernel_name like '%k_node_wgrads%' "
                          "group by kernel_name, counter_name order by kernel_name, counter_name").fetchall()
        for kn, cn, c, v, d in rows:
            out.append(f"| {short(kn)} | {cn} | {c} | {v:.4g} | {d:.0f} |")
            key = "k_block_bwd" if "k_block_bwd" in kn else "k_block_fwd" if "k_block_fwd" in kn else None
            if key and cn in ("FETCH_SIZE", "WRITE_SIZE"):
                traffic.setdefault(key, {})[cn] = v
    out.append("")
    out.append("Reading the counters (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 "
               "FETCH_SIZE reports half of the bytes of a wide coalesced streaming read, so it is doubled.\n")
    ptf = os.path.join(REPO, "profiles", "pmc_traffic.json")
    try:
        pt = json.load(open(ptf))
        if "k_block_bwd" in pt:            # round-2 layout (one workload): move it under its workload key
            pt = {"zinc500k_n64": {k: v for k, v in pt.items() if not k.startswith("_")}}
    except Exception:  # noqa: BLE001
        pt = {}
    pt["_source"] = (f"profiles/{name}_rocprof_summary.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, per workload; "
                     "bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024)")

    def put(wl, tr):
        for k, v in tr.items():
            if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
                pt.setdefault(wl, {})[k] = int(round((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024))
                out.append(f"- `{wl}` `{k}`: HBM traffic per launch = 2 x {v['FETCH_SIZE']:.4g} KB + {v['WRITE_SIZE']:.4g} KB = {pt[wl][k] / 1e6:.1f} MB")
    put("zinc500k_n64", traffic)
    for wl in ("cifar10_n150", "pattern500k_n120_b128"):
        tr = {}
        for sub in (f"pmc_fetch_{wl}", f"pmc_write_{wl}"):
            db = db_of(os.path.join(src, sub))
            if not db:
                continue
            for kn, cn, c, v, d in db.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection "
                                              "where kernel_name like '%k_block_%' or kernel_name like '%k_narrow_%' group by kernel_name, counter_name").fetchall():
                out.append(f"| {wl}: {short(kn)} | {cn} | {c} | {v:.4g} | {d:.0f} |")
                key = "k_block_bwd" if ("k_block_bwd" in kn or "k_narrow_bwd" in kn) else "k_block_fwd" if ("k_block_fwd" in kn or "k_narrow_fwd" in kn) else None
                if key and cn in ("FETCH_SIZE", "WRITE_SIZE"):
                    tr.setdefault(key, {})[cn] = v
        put(wl, tr)
        try:
            line = open(os.path.join(src, f"bench_{wl}.json")).read().strip().splitlines()[-1]
            out.append(f"\n`python bench.py --workload {wl}`:\n\n```json\n{line}\n```\n")
        except Exception:  # noqa: BLE001
            pass
    if len(pt) > 1:
        json.dump(pt, open(ptf, "w"), indent=1)
    try:
        log = open(os.path.join(src, "pytest_gpu.log")).read().strip().splitlines()
        open(os.path.join(REPO, "profiles", f"{name}_pytest_gpu.log"), "w").write("\n".join(log[-12:]) + "\n")
        out.append("\n## pytest -m gpu\n\n```\n" + "\n".join(log[-4:]) + "\n```\n")
    except Exception:  # noqa: BLE001
        pass
    open(os.path.join(REPO, "profiles", f"{name}_rocprof_summary.md"), "w").write("\n".join(out) + "\n")
    print("\n".join(out[-30:]))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""profiles/<name>_narrow_kernels.md from the databases of tools/profile_narrow.sh:
    python tools/prof_narrow_summary.py gpurun_out/<tag> r03_final
Kernel trace + per-launch counter averages of the De = 8 pair kernels on BASELINE config 3 as specified, and the HBM
traffic per launch (FETCH_SIZE / WRITE_SIZE in KB; fetch doubled on gfx950 as MI355X_MICROARCH.md prescribes)."""
import glob, json, os, sqlite3, sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def db_of(d):
    f = sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True))
    return sqlite3.connect(f[0]) if f else None


def tables(db):
    return [r[0] for r in db.execute("select name from sqlite_master where type='table' or type='view'")]


def main():
    src, name = sys.argv[1], sys.argv[2]
    out = [f"# {name} — De = 8 kernels on BASELINE config 3 as specified (cifar10_n150: B=128, N=150, bf16 edge tensors, Ly=4)\n",
           "`tools/profile_narrow.sh`: rocprofv3 --kernel-trace --stats and separate --pmc passes of "
           "`python bench.py --workload cifar10_n150 --steps 10 --warmup 3 --no-cpu-baseline --no-prof`.\n"]
    try:
        b = json.loads(open(os.path.join(src, "bench_cifar.json")).read().strip().splitlines()[-1])
        out.append(f"bench line under rocprof: {b['value']:.0f} graphs/s, {b['ms_per_step']:.3f} ms/step\n")
    except Exception as e:  # noqa: BLE001
        out.append(f"(bench line unreadable: {e})\n")
    db = db_of(os.path.join(src, "kt_cifar"))
    durs = {}
    if db:
        out.append("## kernel trace\n\n| kernel | calls | total_us | avg_us | % |\n|---|---|---|---|---|")
        rows = db.execute("select name, count(*), sum(duration), avg(duration) from kernels group by name order by 3 desc").fetchall()
        tot = sum(r[2] for r in rows)
        for n, c, s, a in rows[:10]:
            out.append(f"| {n[:110]} | {c} | {s / 1e3:.1f} | {a / 1e3:.2f} | {100 * s / tot:.2f} |")
            durs[n] = a
        out.append("")
    vals = {}
    out.append("## counters (per-launch averages)\n\n| kernel | counter | launches | avg value |\n|---|---|---|---|")
    for sub in ("pmc_sq_cifar", "pmc_fetch_cifar", "pmc_write_cifar"):
        db = db_of(os.path.join(src, sub))
        if not db:
            continue
        for k, c, n, v in db.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                                     "where kernel_name like '%k_narrow_%' group by kernel_name, counter_name order by 1, 2"):
            out.append(f"| {k[:60]} | {c} | {n} | {v:.4g} |")
            vals.setdefault(k, {})[c] = v
    out.append("")
    B, N, De, Dh, s = 128, 150, 8, 64, 2   # bf16 edge tensors
    alg = {"bwd": B * (3 * N * N * De * s + 3 * N * Dh * 4), "fwd": B * (2 * N * N * De * s + 3 * N * Dh * 4)}
    out.append("## derived\n")
    for k, v in vals.items():
        kind = "bwd" if "bwd" in k else "fwd"
        fetch = 2 * v.get("FETCH_SIZE", 0) * 1024
        write = v.get("WRITE_SIZE", 0) * 1024
        dur = durs.get(k, 0)
        line = f"- `{k[:50]}`: algorithmic {alg[kind] / 1e6:.0f} MB/launch"
        if dur:
            line += f" / {dur / 1e3:.1f} us = {alg[kind] / dur:.2f} GB/s*1e0 ({alg[kind] / dur / 8000:.3f} of 8 TB/s)"
        if fetch or write:
            line += f"; HBM traffic {(fetch + write) / 1e6:.0f} MB (fetch {fetch / 1e6:.0f} + write {write / 1e6:.0f}) = {(fetch + write) / alg[kind]:.2f}x algorithmic"
        if v.get("SQ_WAVE_CYCLES"):
            line += (f"; waves waiting on an instruction {100 * v.get('SQ_WAIT_INST_ANY', 0) / v['SQ_WAVE_CYCLES']:.0f} % of wave cycles, "
                     f"issuing {100 * v.get('SQ_ACTIVE_INST_ANY', 0) / v['SQ_WAVE_CYCLES']:.0f} %, matrix pipe busy cycles {v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0):.3g}, "
                     f"VALU instructions {v.get('SQ_INSTS_VALU', 0):.3g}")
        out.append(line)
    open(os.path.join(REPO, "profiles", f"{name}_narrow_kernels.md"), "w").write("\n".join(out) + "\n")
    print("\n".join(out[-6:]))


if __name__ == "__main__":
    main()

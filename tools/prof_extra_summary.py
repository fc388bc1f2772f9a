#!/usr/bin/env python3
"""profiles/<name>_mfma_kernels.md from the databases of tools/profile_extra.sh:
    python tools/prof_extra_summary.py gpurun_out/extra r01_final"""
import glob, json, os, sqlite3, sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NSIMD = 1024


def db_of(d):
    f = sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True))
    return sqlite3.connect(f[0]) if f else None


def section(out, src, tag, title, jsonfile, like):
    out.append(f"## {title}\n")
    try:
        out.append("```json\n" + open(os.path.join(src, jsonfile)).read().strip().splitlines()[-1] + "\n```\n")
    except Exception:  # noqa: BLE001
        pass
    db = db_of(os.path.join(src, tag + "_kt"))
    if db:
        out.append("### rocprofv3 --kernel-trace --stats\n\n| kernel | calls | avg_us | % |\n|---|---|---|---|")
        rows = db.execute("select name, count(*), sum(duration), avg(duration) from kernels group by name order by 3 desc").fetchall()
        tot = sum(r[2] for r in rows)
        for n, c, s, a in rows[:8]:
            out.append(f"| {n[:70]} | {c} | {a / 1e3:.1f} | {100 * s / tot:.1f} |")
        out.append("")
    vals = {}
    out.append("### rocprofv3 --pmc (per-launch averages)\n\n| kernel | counter | avg value |\n|---|---|---|")
    for sub in (tag + "_a", tag + "_b"):
        db = db_of(os.path.join(src, sub))
        if not db:
            continue
        for k, c, v in db.execute("select kernel_name, counter_name, avg(value) from counters_collection "
                                  f"where kernel_name like '{like}' group by kernel_name, counter_name order by 1, 2"):
            out.append(f"| {k[:60]} | {c} | {v:.4g} |")
            vals.setdefault(k, {})[c] = v
    out.append("\nDerived (1024 SIMDs; SQ_VALU_MFMA_BUSY_CYCLES in cycles, GRBM_GUI_ACTIVE summed over the 8 XCDs, "
               "SQ_LDS_* in LDS cycles):\n")
    db = db_of(os.path.join(src, tag + "_kt"))
    durs = dict(db.execute("select name, avg(duration) from kernels group by name").fetchall()) if db else {}
    for k, v in vals.items():
        if not v.get("SQ_INSTS_MFMA"):
            continue
        cyc = v.get("GRBM_GUI_ACTIVE", 0) / 8
        busy = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / NSIMD / cyc if cyc else 0
        clk = cyc / durs[k] * 1e3 if k in durs and durs[k] else 0
        conf = v.get("SQ_LDS_BANK_CONFLICT", 0) / v["SQ_LDS_IDX_ACTIVE"] if v.get("SQ_LDS_IDX_ACTIVE") else 0
        out.append(f"- `{k[:60]}`: matrix pipe busy {100 * busy:.0f} % of {cyc:.3g} cycles ({v['SQ_INSTS_MFMA']:.3g} MFMAs), "
                   f"clock {clk:.0f} MHz, LDS bank-conflict share {100 * conf:.0f} %")
    out.append("")


def main():
    src, name = sys.argv[1], sys.argv[2]
    out = [f"# {name} — MFMA-bound kernels (channel FFN, MFMA inner op): rocprofv3 evidence\n",
           "Collected by `tools/profile_extra.sh` (counter passes separate from each other, `--kernel-trace` only); "
           "written by `tools/prof_extra_summary.py`.\n"]
    section(out, src, "ffn", "channel FFN, `tools/bench_ffn.py` ([128,64,64,64] fp32, fwd+bwd)", "ffn.json", "%k_ffn_%")
    section(out, src, "core", "MFMA inner op, `tools/bench_core.py cfg5` (B=8, N=512, d=64, fwd+bwd)", "core.json", "%k_attn_%")
    for extra, title in (("../block_cfg5.json", "config-5 block scope (composed path), `tools/bench_block_cfg5.py [8|32]`"),
                         ("../core_cfg5_b32.json", "MFMA inner op at B=32, `tools/bench_core.py cfg5_b32`"),
                         ("../scope_table.jsonl", "scope table, `tools/scope_table.sh`")):
        p = os.path.join(src, extra)
        if os.path.exists(p):
            body = "\n".join(l for l in open(p).read().splitlines() if l.startswith("{"))
            out.append(f"## {title}\n\n```json\n{body}\n```\n")
    open(os.path.join(REPO, "profiles", f"{name}_mfma_kernels.md"), "w").write("\n".join(out))
    print("\n".join(out[-40:]))


if __name__ == "__main__":
    main()

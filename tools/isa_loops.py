#!/usr/bin/env python3
"""Instruction mix of the loops of one kernel in a gfx950 assembly listing (hipcc -S --cuda-device-only).

Usage: python tools/isa_loops.py file.s kernel-name-substring [min-instructions]
For every backward branch (a loop) of the kernel prints the instruction counts by class between the
branch target and the branch: MFMA, other VALU, transcendental, LDS, VMEM, SALU, waits, barriers.
The issue-cycle estimate uses the measured gfx950 costs (16x16x4 f32 MFMA 32 cycles, VALU 2.5,
transcendental 4: tools/micro/valu_rates.hip); LDS / VMEM / SALU issue beside them."""
import re
import sys


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(("v_exp", "v_rcp", "v_log", "v_sqrt", "v_rsq", "v_sin", "v_cos")):
        return "trans"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, name = sys.argv[1], sys.argv[2]
    minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(rf"^\S*{re.escape(name)}\S*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    labels, insts = {}, []
    for i in range(start, end + 1):
        l = lines[i].split(";")[0].rstrip()
        m = re.match(r"^(\.LBB\S+):", l)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        l = l.strip()
        if not l or l.startswith(".") or l.endswith(":"):
            continue
        insts.append(l)
    print(f"{lines[start].strip()}  {len(insts)} instructions")
    for idx, l in enumerate(insts):
        m = re.match(r"s_cbranch\S*\s+(\.LBB\S+)|s_branch\s+(\.LBB\S+)", l)
        if not m:
            continue
        tgt = labels.get(m.group(1) or m.group(2))
        if tgt is None or tgt > idx or idx - tgt < minlen:
            continue
        cnt = {}
        for x in insts[tgt:idx + 1]:
            c = classify(x.split()[0])
            cnt[c] = cnt.get(c, 0) + 1
        g = lambda k: cnt.get(k, 0)
        cyc = 32 * g("mfma") + 2.5 * g("valu") + 4 * g("trans")
        print(f"  loop [{tgt}:{idx}] len {idx - tgt + 1}: " + " ".join(f"{k}={v}" for k, v in sorted(cnt.items())) +
              f"  | issue cycles ~{cyc:.0f} (mfma share {32 * g('mfma') / max(cyc, 1):.2f})")


if __name__ == "__main__":
    main()

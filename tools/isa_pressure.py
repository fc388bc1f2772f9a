#!/usr/bin/env python3
"""VGPR live-range profile of the largest loop of a kernel in a gfx950 assembly listing (hipcc -S --cuda-device-only).

Usage: python tools/isa_pressure.py file.s kernel-name-substring [every]
Treats the loop body as straight-line code (exec-masked side branches are walked through in program order), runs a backward
liveness pass with the loop's wrap-around, and prints the number of live VGPRs every `every` instructions together with the
scheduling-barrier / MFMA landmarks -- where the register allocator's peak sits and which phase owns it."""
import re
import sys


def regs(tok):
    out = []
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b', tok):
        if m.group(3) is not None:
            out.append(int(m.group(3)))
        else:
            out.extend(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def main():
    path, kern = sys.argv[1], sys.argv[2]
    every = int(sys.argv[3]) if len(sys.argv) > 3 else 25
    L = open(path).read().split('\n')
    start = next(i for i, l in enumerate(L) if l.startswith('_Z') and kern in l and re.match(r'^_Z\w+:', l))
    end = next(i for i in range(start, len(L)) if 's_endpgm' in L[i])
    K = L[start:end]
    labels = {m.group(1): i for i, l in enumerate(K) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
    loops = []
    for i, l in enumerate(K):
        m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    lo, hi = max(loops, key=lambda x: x[1] - x[0])
    body = []
    for l in K[lo:hi + 1]:
        t = l.strip()
        if not t or t.startswith(';') or t.startswith('.') and t.endswith(':'):
            if 'sched_barrier' in t:
                body.append(('fence', [], [], t))
            continue
        t = t.split(';')[0].strip()
        op, _, rest = t.partition(' ')
        ops = [o.strip() for o in rest.split(',')] if rest else []
        if op.startswith(('s_', ';;')):
            body.append((op, [], [], t)); continue
        is_store = op.startswith(('ds_write', 'global_store', 'scratch_store', 'buffer_store', 'global_load_lds'))
        if is_store or not ops:
            d, u = [], [r for o in ops for r in regs(o)]
        else:
            d, u = regs(ops[0]), [r for o in ops[1:] for r in regs(o)]
            if op.startswith('v_mfma') or 'dpp' in t or op.startswith(('v_fmac', 'v_mac', 'v_pk_fma', 'v_cndmask')) and False:
                pass
            if op.startswith(('v_fmac', 'v_mac')) or 'dpp' in t:
                u = u + d          # read-modify-write destinations
        body.append((op, d, u, t))
    n = len(body)
    live = set()
    lv = [None] * n
    for _ in range(2):
        for i in range(n - 1, -1, -1):
            op, d, u, t = body[i]
            live = (live - set(d)) | set(u)
            lv[i] = len(live)
    peak = max(range(n), key=lambda i: lv[i])
    print(f'loop: {n} instructions, peak {lv[peak]} live VGPRs at #{peak}: {body[peak][3][:80]}')
    mf = 0
    for i, (op, d, u, t) in enumerate(body):
        if op.startswith('v_mfma'):
            mf += 1
        if op == 'fence' or i % every == 0 or i == peak:
            print(f'{i:5d} live {lv[i]:4d} mfma# {mf:3d}  {t[:70]}')


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace database:  python tools/gap_stat.py <dir with the .db>
Prints, over the last `tail` fraction of the dispatches (the timed steps), the busy time (sum of kernel durations), the span (first start
to last end) and the gap before each kernel family (end of the previous kernel -> start of this one)."""
import glob, os, sqlite3, sys, collections
d = sys.argv[1]; tail = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
f = sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True))[0]
db = sqlite3.connect(f)
rows = db.execute("select name, start, end from kernels order by start").fetchall()
rows = rows[int(len(rows) * (1 - tail)):]
busy = sum(e - s for _, s, e in rows); span = rows[-1][2] - rows[0][1]
print(f"dispatches {len(rows)}  busy {busy/1e3:.1f} us  span {span/1e3:.1f} us  idle {(span-busy)/1e3:.1f} us = {100*(span-busy)/span:.1f} % ; mean gap {(span-busy)/1e3/(len(rows)-1):.2f} us")
gaps = collections.defaultdict(list)
short = lambda n: n.split("(")[0].replace("void ", "").split("<")[0]
for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
    gaps[(short(n0), short(n1))].append((s1 - e0) / 1e3)
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:14]:
    v.sort()
    print(f"  {k[0]:>22} -> {k[1]:<22} n {len(v):5d}  median {v[len(v)//2]:7.2f} us  mean {sum(v)/len(v):7.2f}  total {sum(v):9.1f}")

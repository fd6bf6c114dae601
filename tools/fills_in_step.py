#!/usr/bin/env python3
"""Fill / copy / elementwise torch kernels of the last profiled step with the kernel that follows each: python tools/fills_in_step.py <rocpd db> [last_seconds]
(which zero-fills and stray torch kernels sit on the step's critical path, and how long they are)."""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select start, end, name from kernels order by start").fetchall()
last = float(sys.argv[2]) if len(sys.argv) > 2 else 1.3
t_end = rows[-1][1]
rows = [r for r in rows if r[0] >= t_end - last * 1e9]
agg = collections.defaultdict(lambda: [0, 0.0])
for i, (s, e, n) in enumerate(rows):
    if n.startswith("void at::native") or "elementwise" in n or "Memset" in n or "fill" in n.lower():
        nxt = rows[i + 1][2][:50] if i + 1 < len(rows) else "-"
        prv = rows[i - 1][2][:50] if i else "-"
        key = (n[:70], round((e - s) / 1e3 / 5) * 5, prv, nxt)
        agg[key][0] += 1
        agg[key][1] += (e - s) / 1e3
tot = sum(v[1] for v in agg.values())
print(f"torch elementwise / fill kernels in the last {last} s: {sum(v[0] for v in agg.values())} launches, {tot/1e3:.2f} ms")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{v[0]:5d} x ~{k[1]:5d} us = {v[1]/1e3:6.2f} ms  {k[0]}\n          after {k[2]}\n          before {k[3]}")

#!/usr/bin/env python3
"""Kernels around the largest idle gaps of a profiled run: python tools/gap_context.py <rocpd db> [n_gaps] [context] [last_seconds]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
ng = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = int(sys.argv[3]) if len(sys.argv) > 3 else 6
rows = db.execute("select start, end, name from kernels order by start").fetchall()
if len(sys.argv) > 4:
    t_end = rows[-1][1]
    rows = [r for r in rows if r[0] >= t_end - float(sys.argv[4]) * 1e9]
gaps, busy_end = [], rows[0][1]
for i in range(1, len(rows)):
    if rows[i][0] - busy_end > 0:
        gaps.append((rows[i][0] - busy_end, i))
    busy_end = max(busy_end, rows[i][1])
for g, i in sorted(gaps, reverse=True)[:ng]:
    print(f"---- gap {g/1e3:.1f} us before kernel #{i} at t={(rows[i][0]-rows[0][0])/1e6:.1f} ms")
    for j in range(max(0, i - ctx), min(len(rows), i + ctx)):
        s, e, n = rows[j]
        print(f"  {'>>' if j == i else '  '} t={(s-rows[0][0])/1e6:9.3f} ms dur {(e-s)/1e3:8.1f} us  {n[:90]}")

#!/usr/bin/env python3
"""Idle gaps of the GPU inside a profiled run: python tools/gpu_gaps.py <rocpd db> [min_gap_us] [last_seconds] -- lists the largest gaps between
consecutive kernels (end of one -> start of the next) with the kernels on both sides, and the total idle time above the threshold."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
thr = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 50e3
rows = db.execute("select start, end, name from kernels order by start").fetchall()
if len(sys.argv) > 3:   # only the last <seconds> of the run (the timed steps)
    t_end = rows[-1][1]
    rows = [r for r in rows if r[0] >= t_end - float(sys.argv[3]) * 1e9]
gaps, busy_end = [], rows[0][1]
for i in range(1, len(rows)):
    s, e, n = rows[i]
    if s - busy_end > thr:
        gaps.append((s - busy_end, rows[i - 1][2][:60], n[:60], (s - rows[0][0]) / 1e6))
    busy_end = max(busy_end, e)
tot = (rows[-1][1] - rows[0][0]) / 1e6
print(f"span {tot:.1f} ms, {len(gaps)} gaps > {thr/1e3:.0f} us totalling {sum(g[0] for g in gaps)/1e6:.1f} ms")
for g in sorted(gaps, reverse=True)[:25]:
    print(f"{g[0]/1e3:9.1f} us at t={g[3]:8.1f} ms | {g[1]}  ->  {g[2]}")

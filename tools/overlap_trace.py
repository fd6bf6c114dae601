#!/usr/bin/env python3
"""Two-stream picture of one SC-GRPO step from a rocprofv3 --kernel-trace database (rocpd): python tools/overlap_trace.py <db> [last_seconds]
Splits the kernels of the last `last_seconds` by HIP stream / queue, finds the decode window (first .. last launch of the rollout's skinny-GEMM kernels), and
reports for every other queue what ran inside that window: per kernel name the launches, summed and average duration, and the length of the UNION of their
intervals (= time the shadow stream kept the GPU busy next to the decode replay)."""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
last = float(sys.argv[2]) if len(sys.argv) > 2 else 1.4
cols = [r[1] for r in db.execute("pragma table_info(kernels)").fetchall()]
qcol = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else None)
print("columns:", cols, "-> grouping by", qcol)
rows = db.execute(f"select start, end, name, {qcol or 0} from kernels order by start").fetchall()
t_end = rows[-1][1]
rows = [r for r in rows if r[0] >= t_end - last * 1e9]
dec = [r for r in rows if "skinny" in r[2]]
d0, d1 = dec[0][0], dec[-1][1]
print(f"decode window {(d1 - d0) / 1e6:.1f} ms, {len(dec)} skinny launches")
if "queue_id" in cols and "stream_id" in cols:
    pairs = db.execute("select stream_id, queue_id, count(*) from kernels group by stream_id, queue_id").fetchall()
    print("(stream_id, hardware queue_id, launches):", pairs)
dq = collections.Counter(r[3] for r in dec).most_common(1)[0][0]
byq = collections.defaultdict(list)
for r in rows:
    if r[0] >= d0 and r[1] <= d1:
        byq[r[3]].append(r)
for q, rs in byq.items():
    iv = sorted((r[0], r[1]) for r in rs)
    union, cur_s, cur_e = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > cur_e:
            union += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    union += cur_e - cur_s
    print(f"\n== queue {q}{' (decode)' if q == dq else ''}: {len(rs)} launches, busy union {union / 1e6:.1f} ms, sum {sum(r[1] - r[0] for r in rs) / 1e6:.1f} ms")
    agg = collections.defaultdict(lambda: [0, 0])
    for s, e, n, _ in rs:
        a = agg[n.replace("void ", "").replace("(anonymous namespace)::", "")[:70]]
        a[0] += 1
        a[1] += e - s
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:18]:
        print(f"  {t / 1e6:9.2f} ms  {c:6d} x {t / c / 1e3:9.1f} us  {n}")

# ---- timeline excerpt: both queues around the start of the 3rd burst of the busiest non-decode queue
others = [q for q in byq if q != dq]
if others:
    sq = max(others, key=lambda q: len(byq[q]))
    srs = sorted(byq[sq])
    bursts = [srs[0]]
    for a_, b_ in zip(srs, srs[1:]):
        if b_[0] - a_[1] > 2e6:
            bursts.append(b_)
    t0 = bursts[min(2, len(bursts) - 1)][0]
    print(f"\n== timeline around t0 = start of a burst on queue {sq} (us relative to t0; D = decode queue, S = shadow queue)")
    ex = sorted(r for r in rows if t0 - 60e3 <= r[0] <= t0 + 40000e3 and r[3] in (dq, sq))
    # compress runs of the same queue: show the first 3 and last 3 launches of every run
    runs, cur = [], []
    for r in ex:
        if cur and cur[-1][3] != r[3]:
            runs.append(cur)
            cur = []
        cur.append(r)
    if cur:
        runs.append(cur)
    ex = []
    for run in runs[:12]:
        ex += run if len(run) <= 8 else run[:3] + [(run[3][0], run[-4][1], f"... {len(run) - 6} more launches ...", run[0][3])] + run[-3:]
    for s, e, n, q in ex[:140]:
        print(f"  {'D' if q == dq else 'S'} {(s - t0) / 1e3:9.1f} .. {(e - t0) / 1e3:9.1f}  ({(e - s) / 1e3:7.1f} us)  {n.replace('void ', '').replace('(anonymous namespace)::', '')[:60]}")

# ---- per burst of the shadow queue: how much of it the decode queue was running
if others:
    srs = sorted(byq[sq])
    bl, cur = [], [srs[0]]
    for a_, b_ in zip(srs, srs[1:]):
        if b_[0] - a_[1] > 2e6:
            bl.append(cur)
            cur = []
        cur.append(b_)
    bl.append(cur)
    drs = sorted(byq[dq])
    print(f"\n== {len(bl)} bursts on queue {sq}: start (ms into the decode window), length, launches | decode queue inside the burst: launches, busy ms")
    for b in bl:
        b0, b1 = b[0][0], max(r[1] for r in b)
        inside = [r for r in drs if r[1] > b0 and r[0] < b1]
        busy = sum(min(r[1], b1) - max(r[0], b0) for r in inside)
        print(f"  {(b0 - d0) / 1e6:8.1f} ms  {(b1 - b0) / 1e6:7.2f} ms  {len(b):5d} | {len(inside):6d}  {busy / 1e6:7.2f} ms   first: {b[0][2][:40]}")

"""Per-kernel summary (calls / total / avg / min / max) from a rocprofv3 rocpd database -- the text tables under profiles/.
usage: python tools/rocpd_summary.py <results.db> [header line ...] > profiles/rNN_x_kernel_stats.txt"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, count(*), sum(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    for h in sys.argv[2:]:
        print("# " + h)
    print(f"# total kernel time {total / 1e6:.1f} ms")
    print(f"{'kernel':<95} {'calls':>7} {'total_ms':>10} {'pct':>6} {'avg_us':>10} {'min_us':>9} {'max_us':>10}")
    for name, n, tot, mn, mx in rows[:60]:
        print(f"{name[:95]:<95} {n:>7} {tot / 1e6:>10.1f} {100 * tot / total:>6.2f} {tot / n / 1e3:>10.1f} {mn / 1e3:>9.1f} {mx / 1e3:>10.1f}")


if __name__ == "__main__":
    main()

"""Summarise kernel durations from a rocprofv3 rocpd (.db) or kernel-trace CSV into a text table.

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [> profiles/r01_x.txt]
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(
        f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
        f"from kernels group by {name_col} order by sum(end-start) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    print(f"{'kernel':<90} {'calls':>7} {'total_ms':>10} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'pct':>6}")
    for n, c, s, a, mn, mx in rows:
        print(f"{n[:90]:<90} {c:>7} {s/1e6:>10.3f} {a/1e3:>10.2f} {mn/1e3:>10.2f} {mx/1e3:>10.2f} {100*s/tot:>6.2f}")


if __name__ == "__main__":
    main(sys.argv[1])

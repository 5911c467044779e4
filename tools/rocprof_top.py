"""Dump the `top_kernels` view of a rocprofv3 results database (rocpd sqlite) as CSV.

usage: python tools/rocprof_top.py gpurun_out/prof/xx_results.db profiles/name.csv
"""
import csv
import sqlite3
import sys


def main(db, out=None, limit=30):
    c = sqlite3.connect(db)
    cur = c.execute("select * from top_kernels")
    cols = [d[0] for d in cur.description]
    rows = cur.fetchall()
    if out:
        with open(out, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(cols)
            w.writerows(rows)
    for r in rows[:limit]:
        name = r[0].replace("(anonymous namespace)::", "")[:70]
        print(f"{name:70s} calls {r[1]:6d} total_us {r[2]/1e3:10.1f} avg_us {r[3]/1e3:9.1f} {r[4]:6.2f}%")


if __name__ == "__main__":
    main(*sys.argv[1:3])

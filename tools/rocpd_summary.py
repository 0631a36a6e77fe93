#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd) results.db: per-kernel calls / total / avg / min / max, like --stats."""
import sqlite3
import sys


def main(path, header=""):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, "
                       "max(end-start)/1e3 from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    if header:
        print(header)
    print(f"{'kernel':62s} {'calls':>8s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>8s} {'max_us':>10s} {'pct':>6s}")
    for r in rows:
        print(f"{r[0][:62]:62s} {r[1]:8d} {r[2]:12.0f} {r[3]:10.2f} {r[4]:8.2f} {r[5]:10.2f} {100 * r[2] / tot:6.2f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")

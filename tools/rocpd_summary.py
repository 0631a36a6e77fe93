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


def counters(path, header=""):
    """per-kernel summary of a --pmc pass: counter, dispatches, avg / min / max of the counter value"""
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value) from counters_collection "
                       "group by kernel_name, counter_name order by 4 desc").fetchall()
    if header:
        print(header)
    print(f"{'kernel':50s} {'counter':12s} {'dispatches':>10s} {'avg':>12s} {'min':>10s} {'max':>12s}")
    for r in rows[:16]:
        print(f"{r[0][:50]:50s} {r[1]:12s} {r[2]:10d} {r[3]:12.1f} {r[4]:10.1f} {r[5]:12.1f}")


def last_dispatches(path, kernel, count):
    """mean counter value over the last `count` dispatches of `kernel` (the bench's timed-with-events leg)"""
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select counter_name, value from counters_collection where kernel_name like ? order by dispatch_id desc limit ?",
                       (f"%{kernel}%", count)).fetchall()
    vals = [r[1] for r in rows]
    print(f"{kernel}: {rows[0][0] if rows else '-'} mean over last {len(vals)} dispatches = {sum(vals) / max(len(vals), 1):.1f} "
          f"(min {min(vals):.1f}, max {max(vals):.1f})")


if __name__ == "__main__":
    if sys.argv[1] == "--last":
        last_dispatches(sys.argv[2], sys.argv[3], int(sys.argv[4]))
    elif sys.argv[1] == "--pmc":
        counters(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "")
    else:
        main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")

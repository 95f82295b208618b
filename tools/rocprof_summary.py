#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (--kernel-trace --stats) into a small text table
that can be committed under profiles/.   usage: rocprof_summary.py results.db [title]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    title = sys.argv[2] if len(sys.argv) > 2 else db
    c = sqlite3.connect(db)
    print("# rocprofv3 --kernel-trace --stats :: %s" % title)
    print("%-90s %6s %14s %14s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, total, avg, pct in c.execute(
            "select name, total_calls, total_duration, average, percentage from top_kernels"):
        print("%-90s %6d %14.0f %14.0f %6.2f%%" % (name[:90], calls, total, avg, pct))
    print()
    print("# per-dispatch resources (first dispatch of each kernel)")
    seen = set()
    q = ("select name, grid_x, workgroup_x, lds_size, scratch_size, vgpr_count, accum_vgpr_count, sgpr_count "
         "from kernels order by start")
    for row in c.execute(q):
        if row[0] in seen:
            continue
        seen.add(row[0])
        print("%-60s grid=%d wg=%d lds=%d scratch=%d vgpr=%d agpr=%d sgpr=%d" % ((row[0][:60],) + row[1:]))


if __name__ == "__main__":
    main()

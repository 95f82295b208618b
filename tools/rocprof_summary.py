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
    print("# per-dispatch resources (first dispatch of each kernel; vgpr/sgpr as rocprofv3's database reports them --")
    print("#  for what the hardware allocates see the code-object table below)")
    seen = set()
    q = ("select name, grid_x, workgroup_x, lds_size, scratch_size, vgpr_count, accum_vgpr_count, sgpr_count "
         "from kernels order by start")
    for row in c.execute(q):
        if row[0] in seen:
            continue
        seen.add(row[0])
        print("%-60s grid=%d wg=%d lds=%d scratch=%d vgpr=%d agpr=%d sgpr=%d" % ((row[0][:60],) + row[1:]))
    # what the hardware allocates: the code-object notes of the shipped library (tools/kernel_resources.py)
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    so = os.path.join(os.path.dirname(here), "squigglekit_amd", "libsquigglekit_hip.so")
    if os.path.exists(so):
        sys.path.insert(0, here)
        import re
        import kernel_resources as kr
        names = {re.sub(r"\(.*$", "", n.replace("(anonymous namespace)::", "").replace("void ", "")) for n in seen}
        print()
        print("# code-object notes of %s (llvm-readelf --notes): registers per lane, spills, LDS, scratch" % os.path.basename(so))
        print("%-70s %5s %5s %7s %7s %7s %8s" % ("kernel", "vgpr", "sgpr", "v_spill", "s_spill", "lds", "scratch"))
        done = set()
        try:
            for co in kr.code_objects(so):
                for k in kr.kernels(co):
                    name = re.sub(r"\(.*\)$", "", kr.demangle(k["name"]).replace("(anonymous namespace)::", "")).replace("void ", "")
                    if name in names and name not in done:
                        done.add(name)
                        print("%-70s %5s %5s %7s %7s %7s %8s" % (name[:70], k.get("vgpr_count"), k.get("sgpr_count"),
                                                                   k.get("vgpr_spill_count"), k.get("sgpr_spill_count"),
                                                                   k.get("group_segment_fixed_size"),
                                                                   k.get("private_segment_fixed_size")))
        except Exception as e:                                       # noqa: BLE001 -- tooling: report and go on
            print("# (code-object notes unavailable: %r)" % e)


if __name__ == "__main__":
    main()

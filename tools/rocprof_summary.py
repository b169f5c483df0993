#!/usr/bin/env python
"""Summarise a rocprofv3 results database (rocpd sqlite, written by `rocprofv3 --kernel-trace --stats`)
into the per-kernel table kept under profiles/.

    python tools/rocprof_summary.py gpurun_out/prof/x_results.db > profiles/r01_xxx_kernel_stats.txt
"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                     "from kernels group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    span = c.execute("select min(start), max(end) from kernels").fetchone()
    print("# rocprofv3 --kernel-trace --stats summary of %s" % path)
    print("# total kernel time %.3f ms over a %.3f ms span, %d dispatches" % (tot / 1e6, (span[1] - span[0]) / 1e6,
                                                                          sum(r[1] for r in rows)))
    print("%-90s %8s %12s %10s %10s %10s %7s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
    for name, n, s, a, mn, mx in rows:
        print("%-90s %8d %12.1f %10.2f %10.2f %10.2f %6.2f%%" % (name[:90], n, s / 1e3, a / 1e3, mn / 1e3, mx / 1e3,
                                                                 100.0 * s / tot))
    regs = c.execute("select distinct name, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size, "
                     "workgroup_x from kernels group by name").fetchall()
    print("\n# resources: kernel, arch_vgpr, accum_vgpr, sgpr, lds_bytes, scratch, workgroup")
    for r in regs:
        print("# %-88s %s" % (r[0][:88], " ".join(str(x) for x in r[1:])))


if __name__ == "__main__":
    main(sys.argv[1])

"""Summarise a rocprofv3 (rocpd sqlite) result: per-kernel calls / total / avg / min / max,
like `--stats` prints.  usage: python tools/rocprof_summary.py results.db [out.txt]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    rows = cur.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, "
                       "max(end-start)/1e3, max(vgpr_count), max(accum_vgpr_count), max(lds_size) "
                       "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    lines = ["# rocprofv3 --kernel-trace --stats summary (durations from the kernel-dispatch table)",
             "%-86s %6s %10s %10s %10s %10s %6s %5s %5s %7s" % ("kernel", "calls", "total_ms", "avg_us", "min_us",
                                                              "max_us", "pct", "vgpr", "agpr", "lds")]
    for r in rows:
        lines.append("%-86s %6d %10.3f %10.1f %10.1f %10.1f %6.2f %5s %5s %7s"
                     % (r[0][:86], r[1], r[2], r[3], r[4], r[5], 100 * r[2] / tot, r[6], r[7], r[8]))
    txt = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt)
    else:
        print(txt)


main()

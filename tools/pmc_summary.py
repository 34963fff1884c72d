"""Per-kernel PMC counter sums/averages from a rocprofv3 --pmc result db.
usage: python tools/pmc_summary.py results.db [out.txt]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    rows = cur.execute("select kernel_name, counter_name, count(*), sum(value), avg(value) from counters_collection "
                       "group by kernel_name, counter_name order by 4 desc").fetchall()
    lines = ["# rocprofv3 --pmc summary: per kernel and counter: dispatches, sum, average per dispatch",
             "%-90s %-28s %7s %16s %16s" % ("kernel", "counter", "n", "sum", "avg")]
    for r in rows:
        lines.append("%-90s %-28s %7d %16.6g %16.6g" % (r[0][:90], r[1], r[2], r[3], r[4]))
    txt = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt)
    else:
        print(txt)


main()

"""Ordered kernel sequence (start offset, duration, idle gap before) of ONE step of the operator-API cycle from a rocprofv3
kernel trace of `tools/opapi_only.py N trace` (steps are separated by torch's spin_kernel).
usage: python tools/opapi_sequence.py results.db [out.txt]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if "spin_kernel" in r[0]]
    a, b = marks[-2] + 1, marks[-1]
    seq = rows[a:b]
    t0 = seq[0][1]
    lines, prev_end, busy = [], seq[0][1], 0.0
    for name, s, e in seq:
        short = name.replace("(anonymous namespace)::", "").replace("void ", "").replace("at::native::", "")[:86]
        lines.append("%9.1f us  %7.1f us  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, short))
        busy += (e - s) / 1e3
        prev_end = e
    span = (seq[-1][2] - t0) / 1e3
    torch_k = [r for r in seq if "at::native" in r[0] or "rocprim" in r[0] or "hipcub" in r[0]]
    lines.append("# %d launches (%d of them torch / rocprim: %.1f us), first kernel start to last kernel end %.1f us, kernel time %.1f us, gaps %.1f us"
                 % (len(seq), len(torch_k), sum((r[2] - r[1]) for r in torch_k) / 1e3, span, busy, span - busy))
    txt = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt)
    print(txt)


main()

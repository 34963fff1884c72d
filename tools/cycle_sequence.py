"""Ordered kernel sequence of the LAST cycle in a rocprofv3 (rocpd sqlite) trace of tools/cycle_only.py:
start offset, duration and idle gap before each kernel.  usage: python tools/cycle_sequence.py results.db [out.txt]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    # the last occurrence of the first kernel of a cycle (k_project_sphere: two per cycle -> take the one before last pair)
    marks = [i for i, r in enumerate(rows) if "k_project_sphere" in r[0]]
    if marks:
        a, b = marks[-4], marks[-2]             # one full cycle: from its first projection to the next cycle's
    else:                                       # SIREN cycle: its first projection packs the weights (k_siren_wscale)
        marks = [i for i, r in enumerate(rows) if "k_siren_wscale" in r[0]]
        a, b = marks[-3], marks[-2]
    seq = rows[a:b]
    t0 = seq[0][1]
    lines, prev_end, busy = [], seq[0][1], 0.0
    for name, s, e in seq:
        short = name.replace("(anonymous namespace)::", "").replace("void ", "")[:70]
        lines.append("%9.1f us  %7.1f us  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, short))
        busy += (e - s) / 1e3
        prev_end = e
    span = (rows[b][1] - t0) / 1e3
    lines.append("# %d launches, span %.1f us, kernel time %.1f us, gaps %.1f us" % (len(seq), span, busy, span - busy))
    txt = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt)
    print(txt)


main()

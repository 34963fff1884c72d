"""What ONE rank of an N-rank run launches in one cycle: the kernels between the marker pairs of
`ISO_TRACE_RANK=r ISO_WORLDS=8 python tools/rank_share_bench.py siren 1000000 1` in a rocprofv3 kernel trace (a marker
before and after each of the rank's segments; the exchanges between segments are emulated outside them).
usage: python tools/rank_sequence.py results.db [out.txt]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    marks = [i for i, r in enumerate(rows) if "spin_kernel" in r[0]]
    assert len(marks) % 2 == 0 and marks, "expected marker pairs"
    lines, busy, n, torch_n, torch_us, seg_t = [], 0.0, 0, 0, 0.0, 0.0
    for k in range(0, len(marks), 2):
        a, b = marks[k] + 1, marks[k + 1]
        seg = rows[a:b]
        if not seg:
            continue
        lines.append("-- segment %d: %d launches, %.1f us from first start to last end" % (k // 2, len(seg), (seg[-1][2] - seg[0][1]) / 1e3))
        seg_t += (seg[-1][2] - seg[0][1]) / 1e3
        prev = seg[0][1]
        for name, s, e in seg:
            short = name.replace("(anonymous namespace)::", "").replace("void ", "").replace("at::native::", "")[:90]
            lines.append("   %7.1f us  gap %6.1f  %s" % ((e - s) / 1e3, (s - prev) / 1e3, short))
            busy += (e - s) / 1e3
            n += 1
            if "at::native" in name or "rocprim" in name or "rocclr" in name:
                torch_n += 1
                torch_us += (e - s) / 1e3
            prev = e
    lines.append("# %d launches (%d torch / runtime copies+fills: %.1f us), kernel time %.1f us, segments' spans %.1f us"
                 % (n, torch_n, torch_us, busy, seg_t))
    txt = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt)
    print(txt)


main()

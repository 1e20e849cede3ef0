"""Kernel timeline of the LAST optimizer step in a rocprofv3 rocpd database: start offset, duration and the idle gap before
each kernel.  python tools/ktimeline.py DIR/NAME_results.db [name of the first kernel of a step, default encode_prep]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    first = sys.argv[2] if len(sys.argv) > 2 else "encode_prep"
    rows = db.execute('select name, start, "end" from kernels order by start').fetchall()
    starts = [i for i, r in enumerate(rows) if first in r[0]]
    if len(starts) < 2:
        print("no step boundary found")
        return
    a, b = starts[-2], starts[-1]
    t0, prev_end, gaps, busy = rows[a][1], None, 0.0, 0.0
    print("# start_us | dur_us | gap_before_us | kernel")
    for name, s, e in rows[a:b]:
        gap = 0.0 if prev_end is None else (s - prev_end) / 1e3
        gaps += max(gap, 0.0)
        busy += (e - s) / 1e3
        print(f"{(s - t0) / 1e3:9.1f} | {(e - s) / 1e3:7.2f} | {gap:6.2f} | {name[:110]}")
        prev_end = e
    print(f"# {b - a} kernels, busy {busy:.1f} us, gaps {gaps:.1f} us, span {(rows[b][1] - t0) / 1e3:.1f} us")


if __name__ == "__main__":
    main()

"""Summarise a rocprofv3 rocpd database (rocprofv3 --kernel-trace --stats -d DIR -o NAME) into the per-kernel table
kept under profiles/:  python tools/kstats.py DIR/NAME_results.db STEPS [header text]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = int(sys.argv[2])
    hdr = sys.argv[3] if len(sys.argv) > 3 else ""
    rows = db.execute('select name, count(*), sum("end" - start), min("end" - start), max("end" - start) '
                      "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    if hdr:
        print("# " + hdr)
    print("# kernel | calls | us/step | avg_us | min_us | max_us | pct")
    for name, n, s, mn, mx in rows:
        print(f"{name[:150]} | {n} | {s / 1e3 / steps:.1f} | {s / 1e3 / n:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100.0 * s / tot:.1f}%")
    print(f"# total kernel time {tot / 1e3 / steps:.1f} us/step over {steps} steps")


if __name__ == "__main__":
    main()

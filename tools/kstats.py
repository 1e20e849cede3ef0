"""Summarise a rocprofv3 rocpd database (rocprofv3 --kernel-trace --stats -d DIR -o NAME) into the per-kernel table
kept under profiles/:  python tools/kstats.py DIR/NAME_results.db STEPS [header text]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    steps = int(sys.argv[2])
    hdr = sys.argv[3] if len(sys.argv) > 3 else ""
    durs = {}
    for name, d in db.execute('select name, "end" - start from kernels'):
        durs.setdefault(name, []).append(d)
    rows = sorted(((n, len(v), sum(v), min(v), max(v), sorted(v)[len(v) // 2]) for n, v in durs.items()), key=lambda r: -r[2])
    tot = sum(r[2] for r in rows)
    if hdr:
        print("# " + hdr)
    # the median is what the kernel costs; the mean also carries the rare preempted / first-touch launch (max column)
    print("# kernel | calls | us/step | avg_us | median_us | min_us | max_us | pct")
    for name, n, s, mn, mx, med in rows:
        print(f"{name[:150]} | {n} | {s / 1e3 / steps:.1f} | {s / 1e3 / n:.2f} | {med / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100.0 * s / tot:.1f}%")
    print(f"# total kernel time {tot / 1e3 / steps:.1f} us/step over {steps} steps")


if __name__ == "__main__":
    main()

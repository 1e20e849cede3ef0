"""Per-kernel means of the PMC counters in a rocprofv3 rocpd database (rocprofv3 --kernel-trace --pmc ... -d DIR -o NAME):
python tools/pmcstats.py DIR/NAME_results.db [substring filter]"""
import sqlite3
import sys
from collections import defaultdict


def main():
    db = sqlite3.connect(sys.argv[1])
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for name, ctr, val in db.execute("select kernel_name, counter_name, value from counters_collection"):
        if filt and filt not in name:
            continue
        a = acc[name][ctr]
        a[0] += val
        a[1] += 1
    for name, ctrs in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", [0, 1])[0]):
        n = max(v[1] for v in ctrs.values())
        print(f"{name[:110]}  (dispatches {n})")
        wc = ctrs.get("SQ_WAVE_CYCLES", [0, 1])
        wcm = wc[0] / max(1, wc[1])
        for c, (s, k) in sorted(ctrs.items()):
            m = s / max(1, k)
            rel = f"  {100 * m / wcm:5.1f}% of WAVE_CYCLES" if wcm and c.startswith("SQ_") and c != "SQ_WAVE_CYCLES" and "INSTS" not in c else ""
            print(f"    {c:28s} {m:16.0f}{rel}")


if __name__ == "__main__":
    main()

"""Idle time in front of each kernel of the step, averaged over the steps of a rocprofv3 rocpd database (tools/ktrace.sh keeps it with
KT_KEEP=1): for every kernel, the time between the latest end of ANY earlier kernel and its own start (0 if something is still running).
   python tools/kgaps.py DIR/NAME_results.db [first kernel of a step]"""
import sqlite3
import sys
from collections import defaultdict


def main():
    db = sqlite3.connect(sys.argv[1])
    first = sys.argv[2] if len(sys.argv) > 2 else "encode_prep"
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    rows = db.execute(f'select name, start, "end"{", " + qcol if qcol else ""} from kernels order by start').fetchall()
    starts = [i for i, r in enumerate(rows) if first in r[0]]
    acc = defaultdict(list)
    for a, b in zip(starts[3:-1], starts[4:]):     # steady steps only
        latest = rows[a][2]
        for r in rows[a + 1:b]:
            acc[(r[0][:70], r[3] if qcol else 0)].append(max(0.0, (r[1] - latest) / 1e3))
            latest = max(latest, r[2])
    print(f"# kernel | {qcol} | mean idle in front of it (us) | steps")
    for (k, q), v in sorted(acc.items(), key=lambda kv: -sum(kv[1]) / len(kv[1])):
        if sum(v) / len(v) >= 0.3:
            print(f"{k} | {q} | {sum(v) / len(v):.2f} | {len(v)}")


if __name__ == "__main__":
    main()

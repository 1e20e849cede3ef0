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
    acc, same = defaultdict(list), defaultdict(list)
    for a, b in zip(starts[3:-1], starts[4:]):     # steady steps only
        latest, last_on = rows[a][2], {}
        for r in rows[a:b]:
            q = r[3] if qcol else 0
            if r is not rows[a]:
                acc[(r[0][:70], q)].append(max(0.0, (r[1] - latest) / 1e3))
                if q in last_on:
                    same[(r[0][:70], q)].append((r[1] - last_on[q]) / 1e3)
            latest = max(latest, r[2])
            last_on[q] = r[2]
    print(f"# kernel | {qcol} | mean idle of the whole device in front of it (us) | mean gap behind the previous kernel of ITS queue (us) | steps")
    for (k, q), v in sorted(acc.items(), key=lambda kv: -sum(same.get(kv[0], [0.0])) / max(1, len(same.get(kv[0], [0.0])))):
        sv = same.get((k, q), [0.0])
        if sum(v) / len(v) >= 0.3 or sum(sv) / len(sv) >= 0.5:
            print(f"{k} | {q} | {sum(v) / len(v):.2f} | {sum(sv) / len(sv):.2f} | {len(v)}")


if __name__ == "__main__":
    main()

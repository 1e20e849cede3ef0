"""Summarise gpurun_out/refresh (written by tools/refresh_profiles.sh on the GPU box) into profiles/:
   rNN_kernel_stats.txt, rNN_hbm_traffic_pmc.txt, rNN_dominant_kernel_traffic.json.
   python tools/make_profiles.py [round tag, default r01]"""
import json
import os
import sqlite3
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "refresh")
DOMINANT = ("strip_kernel<0>", "strip_kernelILi0E")   # strip::strip_kernel<ROLE_YF> (k_score_strip.hip), demangled / mangled spelling
DOMINANT_W = ("strip_kernel<1>", "strip_kernelILi1E")  # strip::strip_kernel<ROLE_W>


def kernel_stats(db_path, steps):
    db = sqlite3.connect(db_path)
    rows = db.execute('select name, count(*), sum("end" - start), min("end" - start), max("end" - start) '
                      "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    out = ["# kernel | calls | us/step | avg_us | min_us | max_us | pct"]
    for name, n, s, mn, mx in rows:
        out.append(f"{name[:150]} | {n} | {s / 1e3 / steps:.1f} | {s / 1e3 / n:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100.0 * s / tot:.1f}%")
    out.append(f"# total kernel time {tot / 1e3 / steps:.1f} us/step over {steps} steps")
    return out


def counter_means(db_path, counter):
    db = sqlite3.connect(db_path)
    acc = defaultdict(lambda: [0.0, 0])
    for name, val in db.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
        acc[name][0] += val
        acc[name][1] += 1
    return {k: v[0] / max(1, v[1]) for k, v in acc.items()}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    prof = os.path.join(ROOT, "profiles")
    steps = 25
    head = [f"# EDGL_BENCH_SPIN_MS=0 rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline   ({steps} optimizer steps, engine path, round {tag})"]
    bl = os.path.join(SRC, "bench_line.json")
    if os.path.exists(bl):
        head.append("# bench line of the same run: " + open(bl).read().strip())
    with open(os.path.join(prof, f"{tag}_kernel_stats.txt"), "w") as f:
        f.write("\n".join(head + kernel_stats(os.path.join(SRC, "ktrace", "k_results.db"), steps)) + "\n")
    rdb = os.path.join(SRC, "recipe", "k_results.db")
    if os.path.exists(rdb):
        rhead = [f"# EDGL_BENCH_SPIN_MS=0 rocprofv3 --kernel-trace --stats -- python bench.py --workload recipe --steps 20 --warmup 5 --no-cpu-baseline   ({steps} optimizer steps, engine path, round {tag})"]
        rbl = os.path.join(SRC, "recipe_bench_line.json")
        if os.path.exists(rbl):
            rhead.append("# bench line of the same run: " + open(rbl).read().strip())
        with open(os.path.join(prof, f"{tag}_recipe_kernel_stats.txt"), "w") as f:
            f.write("\n".join(rhead + kernel_stats(rdb, steps)) + "\n")
    fetch = counter_means(os.path.join(SRC, "fetch", "f_results.db"), "FETCH_SIZE")
    write = counter_means(os.path.join(SRC, "write", "w_results.db"), "WRITE_SIZE")
    lines = ["# rocprofv3 --kernel-trace --pmc FETCH_SIZE  and (separate pass)  --pmc WRITE_SIZE  -- python bench.py --steps 3 --warmup 2",
             "# per-launch averages; units: KiB as reported (x1024 = bytes).  MI355X_MICROARCH.md §HBM: on gfx950 FETCH_SIZE counts 128-B requests",
             "# as 64 B for wide coalesced streams -> the corrected read bytes are 2 x FETCH_SIZE (upper bound; WRITE_SIZE uncalibrated).",
             "# kernel | FETCH_SIZE KiB | corrected read MB | WRITE_SIZE KiB | write MB"]
    dom = domw = None
    for name in sorted(fetch, key=lambda k: -(fetch[k] + write.get(k, 0.0))):
        fk, wk = fetch[name], write.get(name, 0.0)
        if fk + wk < 2000:
            continue
        lines.append(f"{name[:80]} | {fk:.0f} | {2 * fk * 1024 / 1e6:.1f} | {wk:.0f} | {wk * 1024 / 1e6:.1f}")
        if any(d in name for d in DOMINANT):
            dom = (name, fk, wk)
        if any(d in name for d in DOMINANT_W):
            domw = (name, fk, wk)
    with open(os.path.join(prof, f"{tag}_hbm_traffic_pmc.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    if dom:
        with open(os.path.join(prof, f"{tag}_dominant_kernel_traffic.json"), "w") as f:
            d = {"kernel": "strip::strip_kernel<ROLE_YF>", "fetch_size_kib": round(dom[1], 1),
                 "write_size_kib": round(dom[2], 1),
                 "hbm_bytes_per_launch": int(2 * dom[1] * 1024 + dom[2] * 1024),
                 "note": "separate --pmc passes; read side doubled per MI355X_MICROARCH.md (gfx950 FETCH_SIZE correction)"}
            if domw:
                d["role_w"] = {"kernel": "strip::strip_kernel<ROLE_W>", "fetch_size_kib": round(domw[1], 1), "write_size_kib": round(domw[2], 1),
                               "hbm_bytes_per_launch": int(2 * domw[1] * 1024 + domw[2] * 1024)}
            json.dump(d, f, indent=1)
    print("profiles refreshed:", tag)


if __name__ == "__main__":
    main()

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
    durs = {}
    for name, d in db.execute('select name, "end" - start from kernels'):
        durs.setdefault(name, []).append(d)
    rows = sorted(((n, len(v), sum(v), min(v), max(v), sorted(v)[len(v) // 2]) for n, v in durs.items()), key=lambda r: -r[2])
    tot = sum(r[2] for r in rows)
    # the median is what a kernel costs; the mean also carries the rare preempted / first-touch launch (see the max column)
    out = ["# kernel | calls | us/step | avg_us | median_us | min_us | max_us | pct"]
    for name, n, s, mn, mx, med in rows:
        out.append(f"{name[:150]} | {n} | {s / 1e3 / steps:.1f} | {s / 1e3 / n:.2f} | {med / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100.0 * s / tot:.1f}%")
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
    tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
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
    # whole-step HBM bytes: every kernel of the step (also the small ones) x its launches per step
    calls = {}
    kdb = sqlite3.connect(os.path.join(SRC, "ktrace", "k_results.db"))
    for name, n in kdb.execute("select name, count(*) from kernels group by name"):
        calls[name] = n / steps
    rd = sum(2 * fetch[k] * 1024 * calls.get(k, 0.0) for k in fetch)
    wr = sum(write.get(k, 0.0) * 1024 * calls.get(k, 0.0) for k in fetch)
    lines.append(f"# step total (all kernels x launches per step): corrected read {rd / 1e6:.0f} MB + written {wr / 1e6:.0f} MB = {(rd + wr) / 1e6:.0f} MB"
                 f" -> {(rd + wr) / 6.3e12 * 1e6:.0f} us at the 6.3 TB/s a streaming kernel reaches")
    with open(os.path.join(prof, f"{tag}_hbm_traffic_pmc.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")
    with open(os.path.join(prof, f"{tag}_step_hbm_bytes.json"), "w") as f:
        json.dump({"read_bytes_corrected": int(rd), "write_bytes": int(wr), "total_bytes": int(rd + wr),
                   "floor_us_at_6.3TBps": round((rd + wr) / 6.3e12 * 1e6, 1),
                   "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), read side x2 (gfx950 correction), per-kernel means x launches per step"}, f, indent=1)
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

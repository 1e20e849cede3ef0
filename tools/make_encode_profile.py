"""Summarise gpurun_out/encode (tools/profile_encode.sh) into profiles/rNN_encode_hbm.txt / .json: kernel time and HBM bytes of
the K1 kernels at config 3 (|items| = 1M, L = 200, d = 256)."""
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_profiles import ROOT, counter_means  # noqa: E402


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    ids = sys.argv[2] if len(sys.argv) > 2 else "zipf"
    src = os.path.join(ROOT, "gpurun_out", f"encode_{ids}")
    db = sqlite3.connect(os.path.join(src, "ktrace", "k_results.db"))
    times = {n: (c, s / c / 1e3) for n, c, s in db.execute('select name, count(*), sum("end" - start) from kernels group by name')}
    fetch = counter_means(os.path.join(src, "fetch", "f_results.db"), "FETCH_SIZE")
    write = counter_means(os.path.join(src, "write", "w_results.db"), "WRITE_SIZE")
    line = json.loads(open(os.path.join(src, "bench_line.json")).read())
    alg_f = line["config"]["algorithmic_bytes_fwd"]
    out = [f"# rocprofv3 --kernel-trace --stats / --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --workload encode --ids {ids}",
           "# bench line (profiled run): " + json.dumps(line),
           "# FETCH_SIZE / WRITE_SIZE in KiB per launch as reported; read bytes doubled per MI355X_MICROARCH.md (gfx950 wide reads) = upper bound",
           "# kernel | calls | avg_us | FETCH KiB | WRITE KiB | HBM MB (1x read .. 2x read) | GB/s on the measured bytes"]
    summary = {}
    for name, (calls, us) in sorted(times.items(), key=lambda kv: -kv[1][1]):
        if "encode" not in name:
            continue
        f, w = fetch.get(name, 0.0), write.get(name, 0.0)
        lo, hi = (f + w) * 1024 / 1e6, (2 * f + w) * 1024 / 1e6
        out.append(f"{name[:70]} | {calls} | {us:.2f} | {f:.0f} | {w:.0f} | {lo:.1f} .. {hi:.1f} | {lo / us * 1e3:.0f} .. {hi / us * 1e3:.0f}")
        if "encode_fwd" in name:
            summary = {"kernel": "encode_fwd_kernel", "avg_us": round(us, 2), "fetch_kib": round(f, 1), "write_kib": round(w, 1),
                       "hbm_bytes_per_launch_1x_read": int((f + w) * 1024), "hbm_bytes_per_launch_2x_read": int((2 * f + w) * 1024),
                       "algorithmic_bytes": alg_f, "algorithmic_GBps_in_this_profiled_run": round(alg_f / us * 1e-3, 1),
                       "ids": ids, "unique_item_rows": line["config"].get("unique_item_rows"),
                       "note": ("Zipf-distributed ids: most gathered item rows hit L2 / MALL, so the HBM-side bytes are below the "
                                "algorithmic bytes (one table row per token)") if ids == "zipf" else
                               "uniform ids: (almost) every gathered row is a distinct line of the 512 MB table — FETCH should approach the gathered bytes"}
    with open(os.path.join(ROOT, "profiles", f"{tag}_encode_hbm_{ids}.txt"), "w") as fh:
        fh.write("\n".join(out) + "\n")
    with open(os.path.join(ROOT, "profiles", f"{tag}_encode_hbm_{ids}.json"), "w") as fh:
        json.dump(summary, fh, indent=1)
    print("\n".join(out[3:]))


if __name__ == "__main__":
    main()

"""Summarise gpurun_out/mfma (tools/profile_mfma.sh) into profiles/rNN_mfma_valu_util.txt: MFMA-pipe and VALU busy percentages
per kernel of the headline step (rocprofv3 derived counters MfmaUtil, VALUBusy; one pass each)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_profiles import ROOT, counter_means  # noqa: E402


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    src = os.path.join(ROOT, "gpurun_out", "mfma")
    mf = counter_means(os.path.join(src, "mfma", "m_results.db"), "MfmaUtil")
    va = counter_means(os.path.join(src, "valu", "v_results.db"), "VALUBusy")
    out = ["# rocprofv3 --kernel-trace --pmc MfmaUtil   and (separate pass)   --pmc VALUBusy   -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline",
           "# MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * SIMD_NUM) * 100; per-launch averages, headline step (bf16, engine path)",
           "# kernel | MfmaUtil % | VALUBusy %"]
    for name in sorted(set(mf) | set(va), key=lambda k: -(mf.get(k, 0.0))):
        if mf.get(name, 0.0) < 0.5 and va.get(name, 0.0) < 10.0:
            continue
        out.append(f"{name[:110]} | {mf.get(name, 0.0):.1f} | {va.get(name, 0.0):.1f}")
    with open(os.path.join(ROOT, "profiles", f"{tag}_mfma_valu_util.txt"), "w") as f:
        f.write("\n".join(out) + "\n")
    # the scoring and attention families as JSON (bench.py copies it into roofline_attention.pipe_utilisation_pmc)
    fam = {"strip_kernel<0>": "score_strip_rows", "strip_kernelILi0E": "score_strip_rows", "strip_kernel<1>": "score_strip_table",
           "strip_kernelILi1E": "score_strip_table", "bimau_fwd_kernel": "bimau_fwd", "bimau_bwd_sweep1": "bimau_bwd_sweep1",
           "intensity_bwd_kernel": "bimau_bwd_intensity", "bimau_bwd_sweep2": "bimau_bwd_sweep2"}
    js = {}
    for name in set(mf) | set(va):
        for frag, key in fam.items():
            if frag in name:
                js[key] = {"MfmaUtil_pct": round(mf.get(name, 0.0), 1), "VALUBusy_pct": round(va.get(name, 0.0), 1)}
    js["source"] = "rocprofv3 --kernel-trace --pmc MfmaUtil / --pmc VALUBusy (separate passes), tools/profile_mfma.sh"
    with open(os.path.join(ROOT, "profiles", f"{tag}_mfma_valu_util.json"), "w") as f:
        json.dump(js, f, indent=1)
    print("\n".join(out[:14]))


if __name__ == "__main__":
    main()

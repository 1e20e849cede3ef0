"""profiles/rNN_step_pmc.txt (tools/r05_call2.sh / refresh) -> profiles/rNN_valu_issue.json: VALU instructions per launch of the BiMAU
kernels and the issue-time floor they imply.   python tools/make_valu_floor.py r05
floor = waves x VALU instructions per wave x CYC / (256 CUs x 4 SIMDs) / CLOCK: every VALU / transcendental / MFMA instruction of a
wave passes the ONE vector issue port of its SIMD (DESIGN.md rule 35).  CYC = 3.1 shader cycles — the measured issue cost of a
plain VALU instruction at >= 2 waves per SIMD (rule 33; transcendentals cost 9, so the true floor is higher), CLOCK = 2.1 GHz
(measured inside these kernels: 2.1-2.36 GHz)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
CYC, CLOCK, SIMDS = 3.1, 2.1e9, 1024
KEYS = {"bimau_fwd_kernel": "bimau_fwd", "bimau_bwd_sweep1_kernel": "bimau_bwd_sweep1", "intensity_bwd_kernel": "bimau_bwd_intensity",
        "bimau_bwd_sweep2_kernel": "bimau_bwd_sweep2"}
out = {}
for line in open(os.path.join(ROOT, "profiles", f"{tag}_step_pmc.txt")):
    if line.startswith("#") or "|" not in line:
        continue
    f = [x.strip() for x in line.split("|")]
    for pat, key in KEYS.items():
        if pat in f[0] and key not in out:
            waves, insts = int(f[1]), float(f[8])
            out[key] = {"waves": waves, "valu_insts_per_wave": insts, "valu_insts_per_launch": waves * insts,
                        "issue_floor_us": round(waves * insts * CYC / SIMDS / CLOCK * 1e6, 1)}
out["_model"] = {"cycles_per_valu_inst": CYC, "clock_hz": CLOCK, "simds": SIMDS,
                 "source": f"profiles/{tag}_step_pmc.txt (rocprofv3 --pmc SQ_INSTS_VALU, SQ_WAVES; separate pass)",
                 "note": "SQ_INSTS_VALU counts plain VALU, transcendental and MFMA instructions alike; 3.1 cycles is the measured issue cost of "
                         "the cheapest class (DESIGN.md rule 33), so the floor is a lower bound of the kernels' issue time"}
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_valu_issue.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))

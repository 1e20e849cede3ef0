"""Register / spill / LDS summary of every kernel in a gfx950 assembly file produced with `hipcc -save-temps=obj`
(the .amdhsa metadata block): python tools/kregs.py /tmp/<file>-hip-amdgcn-amd-amdhsa-gfx950.s [name filter]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = []
for b in txt.split("- .agpr_count:")[1:]:
    nm = re.search(r"\.name:\s+(\S+)", b).group(1)
    vg = int(re.search(r"\.vgpr_count:\s+(\d+)", b).group(1))
    sp = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", b).group(1))
    ag = int(re.match(r"\s*(\d+)", b).group(1))
    sc = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", b).group(1))
    rows.append((nm, vg, ag, sp, sc))
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
for (nm, vg, ag, sp, sc), d in zip(rows, names):
    d = re.sub(r"^void ", "", re.sub(r"\(.*", "", d.replace("(anonymous namespace)::", "")))
    if flt in d:
        print(f"{d[:100]:100s} vgpr {vg:4d} agpr {ag:3d} spill {sp:4d} scratch {sc}")

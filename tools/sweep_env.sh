#!/bin/bash
# Runs ON THE GPU BOX: the default bench step under one environment switch at a time, interleaved with the default (A/B on ONE box; boxes of
# the pool differ by a few per cent).   bash tools/sweep_env.sh [reps]  -> gpurun_out/sweep/sweep.txt
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/sweep
mkdir -p "$OUT"
cd "$ROOT"
REPS=${1:-3}
CFGS=("base=1" "EDGL_TN_GROUP_TARGET=384" "EDGL_TN_GROUP_TARGET=448" "EDGL_TN_GROUP_TARGET=576" "EDGL_TN_GROUP_TARGET=640" "EDGL_TN_GROUP_TARGET=768"
      "EDGL_SCORE_TARGET=240" "EDGL_SCORE_TARGET=248" "EDGL_SCORE_TARGET=264" "EDGL_SCORE_TARGET=512" "EDGL_SCORE_FTARGET=240" "EDGL_SCORE_FTARGET=248" "EDGL_SCORE_FTARGET=512"
      "EDGL_LABEL_EARLY=0" "EDGL_LABEL_EARLY=1" "EDGL_L2_EARLY=0" "EDGL_TAIL2=0" "EDGL_TAIL2_STAGGER=1" "EDGL_BIMAU_ORDER=1" "EDGL_DROPBITS=0"
      "EDGL_ADAM_EX=0" "EDGL_L2_PARTS=0" "EDGL_LABEL_FUSED=0" "EDGL_PREP_IN_ENCODER=0" "EDGL_XCD_ORDER=0" "EDGL_TPP_FUSED=0" "EDGL_CE_PARTS=0")
: > "$OUT/raw.txt"
for rep in $(seq 1 "$REPS"); do
  for cfg in "${CFGS[@]}"; do
    ms=$(env "$cfg" python bench.py --steps 100 --warmup 20 --no-extras --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | cut -d' ' -f2)
    echo "$cfg $ms" >> "$OUT/raw.txt"
  done
done
python - "$OUT/raw.txt" > "$OUT/sweep.txt" <<'P'
import sys, statistics as st
from collections import defaultdict
d = defaultdict(list)
for line in open(sys.argv[1]):
    k, *v = line.split()
    if v:
        d[k].append(float(v[0]))
base = st.median(d["base=1"])
print(f"# tools/sweep_env.sh: ms per step of the default bench step under one switch at a time (median of {len(d['base=1'])}, one box); base {base:.4f}")
for k, v in d.items():
    print(f"{k:32s} median {st.median(v):.4f}  min {min(v):.4f}  max {max(v):.4f}  vs base {1e3 * (st.median(v) - base):+.1f} us")
P
cat "$OUT/sweep.txt"

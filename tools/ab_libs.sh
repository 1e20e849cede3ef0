#!/bin/bash
# Runs ON THE GPU BOX: the default bench step under each library of ab_libs/ (EDGL_LIB_PATH), interleaved, REPS times; medians.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd "$ROOT"; mkdir -p gpurun_out/ab
REPS=${1:-4}
: > gpurun_out/ab/raw.txt
for rep in $(seq 1 "$REPS"); do
  for lib in ab_libs/*.so; do
    ms=$(EDGL_LIB_PATH=$ROOT/$lib python bench.py --steps 100 --warmup 20 --no-extras --no-cpu-baseline 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1 | cut -d' ' -f2)
    echo "$(basename $lib) $ms" >> gpurun_out/ab/raw.txt
  done
done
python - <<'P'
import statistics as st
from collections import defaultdict
d = defaultdict(list)
for line in open("gpurun_out/ab/raw.txt"):
    k, *v = line.split()
    if v: d[k].append(float(v[0]))
for k, v in d.items():
    print(f"{k:14s} median {st.median(v):.4f}  all {' '.join(f'{x:.4f}' for x in v)}")
P

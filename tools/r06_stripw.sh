#!/bin/bash
# On the GPU box: the wide strip kernels (k_score_stripw.hip) — parity tests, then the scoring pair alone at C = 256 against the
# generic kernels (EDGL_SCORE_STRIPW=0), at the headline's item count and at config 3's (1 M items).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/stripw
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests/test_gpu_score_strip.py -x -q 2>&1 | tail -15
{
for v in 0 1; do
  EDGL_SCORE_STRIPW=$v C=256 python tools/strip_bench.py
  EDGL_SCORE_STRIPW=$v C=256 R=20480 I=1000001 ZERO=0.475 python tools/strip_bench.py
done
} 2>&1 | tee "$OUT/bench.txt"

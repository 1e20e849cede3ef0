#!/bin/bash
# On the GPU box: A/B x 3 of the wide strip scoring kernels inside the engine step (tools/try_shape.py: 10 + N steps of one batch, HIP events).
#   config 3 (1 M items, T = 201, C = 256, masklen 40): EDGL_SCORE_STRIPW = 0 | 1;   the published recipe (C = 512): EDGL_SCORE_STRIPW512 = 0 | 1
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd "$ROOT"
for rep in 1 2 3; do
  for v in 0 1; do
    EDGL_SCORE_STRIPW=$v python tools/try_shape.py num_units=256 num_items=1000000 seqslen=200 masklen=40 steps=20 2>/dev/null | sed "s/^/EDGL_SCORE_STRIPW=$v  /"
  done
done
for rep in 1 2 3; do
  for v in 0 1; do
    EDGL_SCORE_STRIPW512=$v python tools/try_shape.py num_units=512 seqslen=30 masklen=6 num_items=17770 steps=300 2>/dev/null | sed "s/^/EDGL_SCORE_STRIPW512=$v  /"
  done
done
for rep in 1 2 3; do
  for v in 0 1; do
    EDGL_SCORE_STRIPW=$v python tools/try_shape.py num_units=256 steps=300 2>/dev/null | sed "s/^/EDGL_SCORE_STRIPW=$v  /"
  done
done

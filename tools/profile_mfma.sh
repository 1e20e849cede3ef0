#!/bin/bash
# Runs ON THE GPU BOX (gpurun): MFMA-pipe and VALU busy percentages per kernel of the headline step (derived counters MfmaUtil and
# VALUBusy, one pass each, kernel trace only).  Summarised by tools/make_mfma_profile.py.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/mfma
mkdir -p "$OUT"
export EDGL_BENCH_SPIN_MS=0   # per-kernel tables: without the conditioning GEMMs of bench.py
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --pmc MfmaUtil -d "$OUT/mfma" -o m -- $CMD > "$OUT/mfma.log" 2>&1
rocprofv3 --kernel-trace --pmc VALUBusy -d "$OUT/valu" -o v -- $CMD > "$OUT/valu.log" 2>&1
ls "$OUT"/*

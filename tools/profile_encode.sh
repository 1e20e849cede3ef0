#!/bin/bash
# Runs ON THE GPU BOX (gpurun): K1 at config 3 (bench.py --workload encode --ids zipf|uniform) under rocprofv3 — kernel trace, then FETCH_SIZE and
# WRITE_SIZE in separate passes (MI355X_MICROARCH.md).  Summarised by tools/make_encode_profile.py.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
IDS=${1:-zipf}     # zipf (the recipe's id distribution) | uniform (every gathered row is a distinct HBM line)
OUT=$ROOT/gpurun_out/encode_$IDS
mkdir -p "$OUT"
export EDGL_BENCH_SPIN_MS=0   # per-kernel tables: without the conditioning GEMMs of bench.py
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --workload encode --ids $IDS --steps 20 --warmup 5"
rocprofv3 --kernel-trace --stats -d "$OUT/ktrace" -o k -- $CMD > "$OUT/ktrace.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/fetch" -o f -- $CMD > "$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/write" -o w -- $CMD > "$OUT/write.log" 2>&1
grep -h '"metric"' "$OUT/ktrace.log" | tail -1 > "$OUT/bench_line.json"
ls "$OUT"

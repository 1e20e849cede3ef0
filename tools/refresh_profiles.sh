#!/bin/bash
# Runs ON THE GPU BOX (gpurun): kernel-time trace of the default bench command plus the two HBM-traffic PMC passes
# (separate passes: FETCH_SIZE and WRITE_SIZE do not fit one; see MI355X_MICROARCH.md).  Results land in gpurun_out/ and
# are summarised into profiles/ by tools/make_profiles.py.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/refresh
mkdir -p "$OUT"
export EDGL_BENCH_SPIN_MS=0   # per-kernel tables: without the conditioning GEMMs of bench.py
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats -d "$OUT/ktrace" -o k -- $BENCH > "$OUT/ktrace.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/fetch" -o f -- python $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras > "$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d "$OUT/write" -o w -- python $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras > "$OUT/write.log" 2>&1
grep -h '"metric"' "$OUT/ktrace.log" | tail -1 > "$OUT/bench_line.json"
# the published recipe's shape (bench.py --workload recipe): kernel trace only
rocprofv3 --kernel-trace --stats -d "$OUT/recipe" -o k -- python $ROOT/bench.py --workload recipe --steps 20 --warmup 5 --no-cpu-baseline --no-extras > "$OUT/recipe.log" 2>&1
grep -h '"metric"' "$OUT/recipe.log" | tail -1 > "$OUT/recipe_bench_line.json"
ls "$OUT"/*/ 

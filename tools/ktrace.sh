#!/bin/bash
# On the GPU box: kernel-time trace of a short bench run -> per-kernel table on stdout.  usage: bash tools/ktrace.sh [extra bench args]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/kt
rm -rf "$OUT"; mkdir -p "$OUT"
export EDGL_BENCH_SPIN_MS=0   # per-kernel tables: without the conditioning GEMMs of bench.py
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT" -o k -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras "$@" > "$OUT/log.txt" 2>&1
python $ROOT/tools/kstats.py "$OUT/k_results.db" 25 | cut -c1-150 | head -${KT_LINES:-28}
grep -h '"metric"' "$OUT/log.txt" | cut -c1-260
if [ -n "$KT_TIMELINE" ]; then python $ROOT/tools/ktimeline.py "$OUT/k_results.db" "$KT_TIMELINE"; fi
if [ -z "${KT_KEEP:-}" ]; then rm -f "$OUT/k_results.db"; fi

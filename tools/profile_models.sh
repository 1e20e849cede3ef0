#!/bin/bash
# Runs ON THE GPU BOX (gpurun): kernel-time traces of one optimizer step of the regressive models (bench.py --workload
# tgat|tisasrec|ctsma).  Summarised into profiles/ by tools/make_model_profiles.py.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/models
mkdir -p "$OUT"
export EDGL_BENCH_SPIN_MS=0   # per-kernel tables: without the conditioning GEMMs of bench.py
cd /tmp && export TMPDIR=/tmp
for w in tgat tisasrec ctsma; do
  rocprofv3 --kernel-trace --stats -d "$OUT/$w" -o k -- python $ROOT/bench.py --workload $w --steps 10 --warmup 5 > "$OUT/$w.log" 2>&1
  grep -h '"metric"' "$OUT/$w.log" | tail -1 > "$OUT/$w.json"
done
ls "$OUT"

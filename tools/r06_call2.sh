#!/bin/bash
# Runs ON THE GPU BOX (gpurun), round 6 call 2 (final code state): the whole GPU suite, the default bench line, the kernel / HBM tables of
# tools/refresh_profiles.sh, and the config-3 kernel table (bench.py extras row re-run under the tracer is too long: tools/try_config3.py).
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06
mkdir -p "$OUT"
cd "$ROOT"
timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/gpu_tests.log" 2>&1
echo "gpu tests rc=$?"; tail -3 "$OUT/gpu_tests.log"
python bench.py > "$OUT/bench_default.log" 2>&1
grep -h '"metric"' "$OUT/bench_default.log" | tail -1 > "$OUT/r06_bench_line.json"
cut -c1-300 "$OUT/r06_bench_line.json"
bash tools/refresh_profiles.sh > "$OUT/refresh.log" 2>&1
python tools/make_profiles.py r06 >> "$OUT/refresh.log" 2>&1
cp gpurun_out/refresh/bench_line.json "$OUT/r06_bench_line_profiled.json" 2>/dev/null
cp profiles/r06_kernel_stats.txt profiles/r06_recipe_kernel_stats.txt profiles/r06_hbm_traffic_pmc.txt profiles/r06_step_hbm_bytes.json profiles/r06_dominant_kernel_traffic.json "$OUT"/ 2>/dev/null
rm -rf gpurun_out/refresh/ktrace gpurun_out/refresh/fetch gpurun_out/refresh/write gpurun_out/refresh/recipe
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/c3 -o k -- python $ROOT/tools/try_config3.py > "$OUT/config3.log" 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- python tools/try_config3.py  (BASELINE.json configs[2]: 1 M items, T = 201, C = 256, batch 512, masklen 40; 6 autograd + 11 engine steps and one eval batch: per-launch averages)";
  grep -E "engine ms|autograd ms" "$OUT/config3.log";
  python $ROOT/tools/kstats.py $(find /tmp/c3 -name '*.db' | head -1) 1 | cut -c1-200 | head -30; } > "$OUT/r06_config3_kernel_stats.txt" 2>&1
ls -la "$OUT" | head -40

#!/bin/bash
# Runs ON THE GPU BOX (gpurun), round 6 call 3: the remaining tables of the round on the final kernels — MFMA / VALU utilisation of the
# headline step, the evaluation step's kernel table, SQ counters of the C = 512 strip passes, the config-3 evaluation step.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06c
mkdir -p "$OUT"
cd "$ROOT"
bash tools/profile_mfma.sh > "$OUT/mfma.log" 2>&1
python tools/make_mfma_profile.py r06 >> "$OUT/mfma.log" 2>&1
cp profiles/r06_mfma_valu_util.* "$OUT"/ 2>/dev/null
rm -rf gpurun_out/mfma/mfma gpurun_out/mfma/valu
{ echo "# EDGL_BENCH_EVAL_ROWS=1 KT_LINES=30 bash tools/ktrace.sh --workload eval   (round 6, final kernels)"; EDGL_BENCH_EVAL_ROWS=1 KT_LINES=30 bash tools/ktrace.sh --workload eval; } > "$OUT/r06_eval_kernel_stats.txt" 2>&1
C=512 R=3072 I=17771 ZERO=0.4 bash tools/stripw_pmc.sh r06c/s5 > /dev/null 2>&1
{ echo "# tools/stripw_pmc.sh (C=512 R=3072 I=17771 ZERO=0.4: the published recipe's scoring shape, 1 844 weighted rows): SQ counters of stripw5_kernel (two channel halves) + kernel times"; cat "$OUT/s5/pmc.txt" "$OUT/s5/kstats.txt" | cut -c1-200; } > "$OUT/r06_stripw5_pmc.txt"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/ev3 -o k -- python $ROOT/tools/try_eval3.py > "$OUT/eval3.log" 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- python tools/try_eval3.py   (BASELINE.json configs[2] evaluation step: 1 M items, T = 201, C = 256, 512 sequences; 30 steps: per-step averages)"; grep "eval step ms" "$OUT/eval3.log"; python $ROOT/tools/kstats.py $(find /tmp/ev3 -name '*.db' | head -1) 30 | cut -c1-170 | head -20; } > "$OUT/r06_eval_config3_kernel_stats.txt" 2>&1
ls -la "$OUT"

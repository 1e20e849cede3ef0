#!/bin/bash
# Runs ON THE GPU BOX (gpurun), round 6 call 1: the whole GPU suite, the default bench line, and every profile table of the round
# (kernel trace + HBM PMC passes of tools/refresh_profiles.sh, three SQ passes over the step, the block tail's two forms side by side).
# Everything judged is copied to gpurun_out/r06/ (-> profiles/r06_* here); the rocprof databases stay on the box.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06
mkdir -p "$OUT"
cd "$ROOT"
if [ -z "${SKIP_TESTS:-}" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/gpu_tests.log" 2>&1
  echo "gpu tests rc=$?"; tail -4 "$OUT/gpu_tests.log"
fi
python bench.py > "$OUT/bench_default.log" 2>&1
grep -h '"metric"' "$OUT/bench_default.log" | tail -1 > "$OUT/r06_bench_line.json"
cut -c1-400 "$OUT/r06_bench_line.json"
bash tools/refresh_profiles.sh > "$OUT/refresh.log" 2>&1
python tools/make_profiles.py r06 >> "$OUT/refresh.log" 2>&1
cp gpurun_out/refresh/bench_line.json "$OUT/r06_bench_line_profiled.json" 2>/dev/null
KT_LINES=40 KT_TIMELINE=encode_prep bash tools/ktrace.sh > "$OUT/r06_step_timeline.txt" 2>&1
# SQ passes over the step
export EDGL_BENCH_SPIN_MS=0
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d "$OUT/pmc1" -o p -- $CMD > "$OUT/pmc1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT -d "$OUT/pmc2" -o p -- $CMD > "$OUT/pmc2.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVES SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT -d "$OUT/pmc3" -o p -- $CMD > "$OUT/pmc3.log" 2>&1
cd "$ROOT"
db() { find "$OUT/pmc$1" -name '*.db' | head -1; }
python tools/make_step_pmc.py "$(db 1)" "$(db 2)" "$(db 3)" r06 > profiles/r06_step_pmc.txt 2> "$OUT/step_pmc.err"
python tools/make_valu_floor.py r06 > "$OUT/valu_floor.log" 2>&1
rm -rf "$OUT"/pmc1 "$OUT"/pmc2 "$OUT"/pmc3
bash tools/tail_pmc.sh r06/tailpmc > "$OUT/tail_pmc.log" 2>&1
{ echo "# tools/tail_pmc.sh: SQ counters of the fused block-tail forward alone (tools/tail_probe.py 128 8 512), one workgroup per CU (EDGL_TAIL2=0) against two (EDGL_TAIL2=1)";
  cat "$OUT/tailpmc/tail2_0.txt" "$OUT/tailpmc/tail2_1.txt"; } > profiles/r06_tail2_pmc.txt 2>/dev/null
cp profiles/r06_* "$OUT"/ 2>/dev/null
rm -rf gpurun_out/refresh/ktrace gpurun_out/refresh/fetch gpurun_out/refresh/write gpurun_out/refresh/recipe
ls -la "$OUT"

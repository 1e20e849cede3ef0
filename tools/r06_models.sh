#!/bin/bash
# On the GPU box: the regressive models of config 5 (TGAT / TiSASRec / CTSMA) — kernel tables of one optimizer step (tools/profile_models.sh)
# and one SQ counter pass each (VALU instructions per launch -> issue floor of the interval-attention kernels K11 / K11b).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06m
mkdir -p "$OUT"
bash $ROOT/tools/profile_models.sh > "$OUT/trace.log" 2>&1
cd "$ROOT" && python tools/make_model_profiles.py r06 >> "$OUT/trace.log" 2>&1
cp profiles/r06_*_kernel_stats.txt "$OUT"/ 2>/dev/null
export EDGL_BENCH_SPIN_MS=0
cd /tmp && export TMPDIR=/tmp
for w in tgat tisasrec ctsma; do
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d "$OUT/p_$w" -o p -- python $ROOT/bench.py --workload $w --steps 3 --warmup 2 > "$OUT/p_$w.log" 2>&1
  { echo "### $w"; python $ROOT/tools/pmcstats.py $(find "$OUT/p_$w" -name '*.db' | head -1) attn; } > "$OUT/pmc_$w.txt" 2>&1
  rm -rf "$OUT/p_$w"
done
rm -rf $ROOT/gpurun_out/models/*/
cat "$OUT"/pmc_*.txt | head -150

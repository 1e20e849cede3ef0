#!/bin/bash
# On the GPU box: SQ counters of the wide strip kernels alone (tools/strip_bench.py at C = 256), three passes of 8 counters.
#   bash tools/stripw_pmc.sh [out dir under gpurun_out] ; env R / I / C as tools/strip_bench.py
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/${1:-stripwpmc}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export C=${C:-256} R=${R:-20480} I=${I:-200001}
CMD="python $ROOT/tools/strip_bench.py"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d "$OUT/a" -o p -- $CMD > "$OUT/a.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT -d "$OUT/b" -o p -- $CMD > "$OUT/b.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVES SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT -d "$OUT/c" -o p -- $CMD > "$OUT/c.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$OUT/k" -o k -- $CMD > "$OUT/k.log" 2>&1
for p in a b c; do
  echo "### pass $p"
  python $ROOT/tools/pmcstats.py $(find "$OUT/$p" -name '*.db' | head -1) strip
done > "$OUT/pmc.txt" 2>&1
python $ROOT/tools/kstats.py $(find "$OUT/k" -name '*.db' | head -1) 1 | cut -c1-160 | head -12 > "$OUT/kstats.txt" 2>&1
rm -rf "$OUT"/a "$OUT"/b "$OUT"/c "$OUT"/k
cat "$OUT/pmc.txt" "$OUT/kstats.txt"

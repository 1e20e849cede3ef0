#!/bin/bash
# On the GPU box: SQ counters of the fused block-tail kernels alone (tools/tail_probe.py), one-per-CU (EDGL_TAIL2=0) against two-per-CU form.
#   bash tools/tail_pmc.sh [out dir under gpurun_out]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/${1:-tailpmc}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  export EDGL_TAIL2=$v
  CMD="python $ROOT/tools/tail_probe.py 128 8 512"
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d "$OUT/a$v" -o p -- $CMD > "$OUT/a$v.log" 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT -d "$OUT/b$v" -o p -- $CMD > "$OUT/b$v.log" 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVES SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT -d "$OUT/c$v" -o p -- $CMD > "$OUT/c$v.log" 2>&1
  for p in a b c; do
    echo "### EDGL_TAIL2=$v pass $p"
    python $ROOT/tools/pmcstats.py $(find "$OUT/$p$v" -name '*.db' | head -1) tail
  done > "$OUT/tail2_$v.txt" 2>&1
  rm -rf "$OUT"/a$v "$OUT"/b$v "$OUT"/c$v
done
cat "$OUT/tail2_0.txt" "$OUT/tail2_1.txt"

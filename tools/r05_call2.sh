#!/bin/bash
# Runs ON THE GPU BOX (gpurun), round 5: (1) A/B of the weight-gradient overlap switch and of the key-tile skip inside the real step,
# (2) the engine tests under the overlap switch, (3) phase stamps of the fused tail kernels, (4) PMC passes of the tail kernels.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/c2
mkdir -p "$OUT"
cd "$ROOT"
line() {  # label: one bench run -> ms_per_step / median / attention / dominant kernel
  python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); a=d.get('roofline_attention',{})
        print('$1', 'ms_per_step', d['ms_per_step'], 'median', d['step_ms_hipevents']['median'], 'att_fwd', a.get('forward',{}).get('avg_ms'), 'att_bwd', a.get('backward',{}).get('avg_ms'), 'dom', d['roofline']['avg_launch_ms'], 'loss', d['loss'])
"
}
B="python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-extras"
{
for rep in 1 2 3; do
  for v in 0 1 2 3; do EDGL_DW_OVERLAP=$v $B 2>/dev/null | line "DW_OVERLAP=$v"; done
done
for rep in 1 2 3; do
  for v in 1 0; do EDGL_BIMAU_SKIP=$v $B 2>/dev/null | line "BIMAU_SKIP=$v"; done
done
} > "$OUT/ab.txt" 2>&1
cat "$OUT/ab.txt"
EDGL_DW_OVERLAP=3 timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_headline_parity.py tests/test_gpu_distributed.py -x -q > "$OUT/tests_overlap.log" 2>&1
tail -3 "$OUT/tests_overlap.log"
# timeline with the overlap on
EDGL_DW_OVERLAP=3 KT_LINES=12 KT_TIMELINE=encode_prep bash tools/ktrace.sh > "$OUT/timeline_overlap3.txt" 2>&1
# (3) tail phases: the variants replace the library in place — keep the real one
cp easydgl_amd/libeasydgl_hip.so /tmp/lib_real.so
bash tools/build_phase_variant.sh k_tail > "$OUT/build_fwd.log" 2>&1 && timeout 300 python tools/phase_probe_tail.py tools/variants/lib_phase_k_tail.so > "$OUT/tail_phases_fwd.txt" 2>&1
EXTRA=-DEDGL_PHASE_BWD TAG=_bwd bash tools/build_phase_variant.sh k_tail > "$OUT/build_bwd.log" 2>&1 && timeout 300 python tools/phase_probe_tail.py tools/variants/lib_phase_k_tail_bwd.so bwd > "$OUT/tail_phases_bwd.txt" 2>&1
cp /tmp/lib_real.so easydgl_amd/libeasydgl_hip.so
cat "$OUT/tail_phases_fwd.txt" "$OUT/tail_phases_bwd.txt"
# (4) PMC: where the tail kernels' wave cycles go (SQ: 8 slots per pass)
export EDGL_BENCH_SPIN_MS=0
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d "$OUT/pmc1" -o p -- $CMD > "$OUT/pmc1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT -d "$OUT/pmc2" -o p -- $CMD > "$OUT/pmc2.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVES SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT -d "$OUT/pmc3" -o p -- $CMD > "$OUT/pmc3.log" 2>&1
cd "$ROOT"
for i in 1 2 3; do python tools/pmcstats.py "$OUT/pmc$i"/*/p_results.db tail_ > "$OUT/pmc${i}_tail.txt" 2>&1 || python tools/pmcstats.py $(find "$OUT/pmc$i" -name '*.db' | head -1) tail_ > "$OUT/pmc${i}_tail.txt" 2>&1; done
for i in 1 2 3; do python tools/pmcstats.py $(find "$OUT/pmc$i" -name '*.db' | head -1) > "$OUT/pmc${i}_all.txt" 2>&1; done
head -40 "$OUT/pmc1_tail.txt"
rm -rf "$OUT"/pmc*/  # the databases are large; the text summaries stay

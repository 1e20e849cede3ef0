#!/bin/bash
cd $GRAFT_REPO_ROOT
export EDGL_LIB_PATH=$GRAFT_REPO_ROOT/tools/variants/lib_kb.so
python -m pytest tests/test_gpu_ops.py tests/test_gpu_coding.py -x -q -m gpu -k "bimau or keep_bits or dropout" 2>&1 | tail -2
for v in base kb base kb; do
  if [ $v = base ]; then unset EDGL_LIB_PATH; else export EDGL_LIB_PATH=$GRAFT_REPO_ROOT/tools/variants/lib_$v.so; fi
  KT_LINES=40 bash tools/ktrace.sh > gpurun_out/kb_$v.txt 2>&1
  echo "== $v"; grep -E "bimau_fwd|sweep" gpurun_out/kb_$v.txt | cut -c1-40,95-150
  python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$v', j['ms_per_step'], j['step_ms_hipevents']['median'])"
done

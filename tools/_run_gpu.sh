#!/bin/bash
cd $GRAFT_REPO_ROOT
export EDGL_LIB_PATH=$GRAFT_REPO_ROOT/tools/variants/lib_twb4.so
python -m pytest tests/test_gpu_engine.py -x -q -m gpu -k tail 2>&1 | tail -2
for v in tw twb4 twb7 tw twb4 twb7; do
  export EDGL_LIB_PATH=$GRAFT_REPO_ROOT/tools/variants/lib_$v.so
  KT_LINES=12 bash tools/ktrace.sh > gpurun_out/tw_$v.txt 2>&1
  echo "== $v"; grep -E "tail_fwd|tail_bwd" gpurun_out/tw_$v.txt | cut -c1-40,80-150
done

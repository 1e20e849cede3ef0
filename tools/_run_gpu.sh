#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in rows2 rows2nj2; do
export EDGL_LIB_PATH=$GRAFT_REPO_ROOT/tools/variants/lib_$v.so
python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "bimau" 2>&1 | tail -2
done
for v in rows2 rows2nj2; do
  export EDGL_LIB_PATH=$GRAFT_REPO_ROOT/tools/variants/lib_$v.so
  KT_LINES=40 bash tools/ktrace.sh --workload recipe > gpurun_out/rc_$v.txt 2>&1
  echo "== $v"; grep -E "intensity" gpurun_out/rc_$v.txt | cut -c1-50,90-175; grep -o '"ms_per_step": [0-9.]*' gpurun_out/rc_$v.txt | head -1
done

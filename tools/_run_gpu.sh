#!/bin/bash
cd $GRAFT_REPO_ROOT
for L in none bk32 bk32w3; do
if [ $L = none ]; then unset EDGL_LIB_PATH; else export EDGL_LIB_PATH=$GRAFT_REPO_ROOT/variants/lib_$L.so; fi
echo "== $L"; python -m pytest tests/test_gpu_ops.py -x -q -k "gemm" 2>&1 | tail -1
KT_LINES=16 bash tools/ktrace.sh | cut -c1-150 | grep -i "tile_nn"
done

#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py -x -q -k "gemm" 2>&1 | tail -2
KT_LINES=16 bash tools/ktrace.sh | cut -c1-150 | grep -i "tile_nn\|metric"

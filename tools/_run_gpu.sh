#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py -x -q -k "gemm or deferred or linear" 2>&1 | tail -2
python -m pytest tests/test_gpu_model.py tests/test_gpu_ctsma.py -x -q 2>&1 | tail -2
for X in 0 1; do echo "XCD=$X"; EDGL_XCD_ORDER=$X KT_LINES=9 bash tools/ktrace.sh --workload recipe | grep -E "tn_gemm|tile_nn|metric" | cut -c1-250; done

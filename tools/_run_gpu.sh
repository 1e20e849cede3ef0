#!/bin/bash
cd $GRAFT_REPO_ROOT
export EDGL_LIB_PATH=$GRAFT_REPO_ROOT/variants/lib_xl.so
python -m pytest tests/test_gpu_ops.py tests/test_gpu_coding.py -x -q -k "bimau or mau or attention" 2>&1 | tail -2
EDGL_LABEL_EARLY=1 KT_LINES=14 bash tools/ktrace.sh | cut -c1-150 | grep -i "sweep2\|metric"
unset EDGL_LIB_PATH
EDGL_LABEL_EARLY=1 KT_LINES=14 bash tools/ktrace.sh | cut -c1-150 | grep -i "sweep2"

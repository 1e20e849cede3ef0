#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py -x -q -k "bimau or mau" 2>&1 | tail -2
python -m pytest tests/test_gpu_engine.py tests/test_gpu_sizes.py tests/test_gpu_ctsma.py tests/test_gpu_coding.py tests/test_gpu_model.py -x -q 2>&1 | tail -2
EDGL_LABEL_EARLY=1 KT_LINES=9 bash tools/ktrace.sh | cut -c1-150 | grep -i "intens\|bimau\|metric"

#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py tests/test_gpu_coding.py -x -q -k "bimau or mau or attention" 2>&1 | tail -2
EDGL_LABEL_EARLY=1 KT_LINES=14 bash tools/ktrace.sh | cut -c1-150 | grep -i "bimau_fwd\|intens\|metric"

#!/bin/bash
# scratch driver for one gpurun call (edited per experiment): tests of the touched family, then the kernel table of a short bench run
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py tests/test_gpu_coding.py -x -q -k "bimau or mau or attention" 2>&1 | tail -2
KT_LINES=14 bash tools/ktrace.sh | cut -c1-150

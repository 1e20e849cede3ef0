#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py -x -q -k "encode" 2>&1 | tail -3
python -m pytest tests/test_gpu_model.py tests/test_gpu_engine.py tests/test_gpu_sizes.py -x -q 2>&1 | tail -2
KT_LINES=12 bash tools/ktrace.sh --workload recipe | cut -c1-170

#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py -x -q -k "deferred or gemm" 2>&1 | tail -3
python -m pytest tests/test_gpu_model.py tests/test_gpu_engine.py tests/test_gpu_headline_parity.py -x -q 2>&1 | tail -2
KT_LINES=8 bash tools/ktrace.sh --workload recipe | cut -c1-170
KT_LINES=1 KT_TIMELINE=step_begin bash tools/ktrace.sh | grep -E "reduce|span|metric" | cut -c1-200

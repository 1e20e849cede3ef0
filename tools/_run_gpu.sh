#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py -x -q -k "tpp or zero_fills" 2>&1 | tail -3
python -m pytest tests/test_gpu_engine.py -x -q 2>&1 | tail -2
python bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | cut -c1-330
KT_LINES=1 KT_TIMELINE=step_begin bash tools/ktrace.sh | grep -E "tpp|bimau_fwd|span" | cut -c1-120

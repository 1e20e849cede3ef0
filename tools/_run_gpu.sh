#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_score_strip.py -x -q 2>&1 | tail -4
python -m pytest tests/test_gpu_engine.py tests/test_gpu_headline_parity.py tests/test_gpu_distributed.py tests/test_gpu_ops.py -x -q 2>&1 | tail -3
python bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | cut -c1-330
python bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | cut -c1-330

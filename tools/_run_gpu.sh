#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_score_strip.py tests/test_gpu_engine.py tests/test_gpu_headline_parity.py -x -q 2>&1 | tail -2
for i in 1 2; do python bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['step_ms_hipevents']['median'])"; done
KT_LINES=30 bash tools/ktrace.sh | grep -E "flash_finish" | cut -c1-200

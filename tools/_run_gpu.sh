#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_score_strip.py tests/test_gpu_engine.py tests/test_gpu_headline_parity.py -x -q 2>&1 | tail -2
KT_LINES=30 bash tools/ktrace.sh | cut -c1-150 | grep -i "strip_kernel\|flash_finish\|metric"
for i in 1 2; do python bench.py --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['step_ms_hipevents']['median'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"; done

#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_engine.py tests/test_gpu_headline_parity.py tests/test_gpu_score_strip.py tests/test_gpu_sizes.py tests/test_cpu_host.py -x -q 2>&1 | tail -5 > gpurun_out/pytest_gpu.txt
cat gpurun_out/pytest_gpu.txt
for r in 1 2 3; do
for v in 1 0; do
  EDGL_LABEL_FUSED=$v python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('label_fused=$v', j['ms_per_step'], j['step_ms_hipevents']['median'], j['roofline_attention']['backward']['avg_ms'])"
done
done

#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py tests/test_gpu_ctsma.py tests/test_gpu_engine.py -x -q -k "bimau or ctsma or forward_loss or engine" 2>&1 | tail -6 > gpurun_out/pytest_gpu.txt
cat gpurun_out/pytest_gpu.txt
python bench.py --no-cpu-baseline > gpurun_out/bench_default.log 2>gpurun_out/bench_default.err
tail -1 gpurun_out/bench_default.log | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print(j['ms_per_step'], j['step_ms_hipevents']['median'], j['roofline']['frac'])
for k,v in j.get('extras',{}).items(): print(k, (v if not isinstance(v,dict) else {a:b for a,b in v.items() if a in ('ms_per_step','ms_median')}))
"

#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -5 > gpurun_out/pytest_gpu.txt
cat gpurun_out/pytest_gpu.txt
bash tools/refresh_profiles.sh > gpurun_out/refresh.log 2>&1
bash tools/profile_mfma.sh > gpurun_out/mfma.log 2>&1
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/bench_default.log 2> gpurun_out/bench_default.err
tail -1 gpurun_out/bench_default.log | cut -c1-600

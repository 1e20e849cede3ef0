#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/refresh_profiles.sh > gpurun_out/refresh.log 2>&1
bash tools/profile_mfma.sh > gpurun_out/mfma.log 2>&1
python bench.py > gpurun_out/bench_default.log 2>&1
tail -1 gpurun_out/bench_default.log | cut -c1-250
python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python __graft_entry__.py smoke 2>&1 | tail -2

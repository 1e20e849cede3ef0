#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/refresh_profiles.sh > gpurun_out/refresh.log 2>&1
bash tools/profile_mfma.sh > gpurun_out/mfma.log 2>&1
KT_LINES=45 KT_TIMELINE=encode_fwd bash tools/ktrace.sh > gpurun_out/kt_timeline.txt 2>&1
python bench.py > gpurun_out/bench_default.log 2>gpurun_out/bench_default.err
tail -1 gpurun_out/bench_default.log | cut -c1-300

#!/bin/bash
cd $GRAFT_REPO_ROOT
EDGL_TEST_DUMP=$GRAFT_REPO_ROOT/gpurun_out/tol python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/pytest_gpu.txt
cat gpurun_out/pytest_gpu.txt
python bench.py > gpurun_out/bench_default.log 2>gpurun_out/bench_default.err
tail -1 gpurun_out/bench_default.log | cut -c1-400
tail -3 gpurun_out/bench_default.err

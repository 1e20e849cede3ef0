#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_engine.py -x -q 2>&1 | tail -12 > gpurun_out/pytest_gpu.txt
cat gpurun_out/pytest_gpu.txt

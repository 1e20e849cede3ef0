#!/bin/bash
cd $GRAFT_REPO_ROOT
python tests/metric_proxy_long.py hip --out gpurun_out/r03_metric_proxy_long.json 2>&1 | tail -3
python -m pytest tests/test_gpu_metric_proxy_long.py -x -q -s 2>&1 | tail -4

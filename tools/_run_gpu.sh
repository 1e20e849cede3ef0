#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py -x -q -k "bimau or mau" 2>&1 | tail -2
python -m pytest tests/test_gpu_engine.py tests/test_gpu_ctsma.py -x -q 2>&1 | tail -2
python bench.py --workload recipe --steps 100 --warmup 20 --no-cpu-baseline --no-extras 2>&1 | tail -1 | cut -c1-330
KT_LINES=26 bash tools/ktrace.sh --workload recipe | grep -E "compact_scan|rows_big|metric" | cut -c1-170
python bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | cut -c1-330

#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py -x -q -k "gemm or deferred" 2>&1 | tail -2
python -m pytest tests/test_gpu_engine.py tests/test_gpu_headline_parity.py -x -q 2>&1 | tail -2
for i in 1 2 3; do for X in 0 1; do
echo -n "XCD=$X "; EDGL_XCD_ORDER=$X python bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['step_ms_hipevents']['median'])"
done; done
for X in 0 1; do EDGL_XCD_ORDER=$X KT_LINES=16 bash tools/ktrace.sh | grep -E "tile_nn|tn_gemm" | cut -c1-150; done

#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2; do
for L in cur allsb; do
if [ $L = allsb ]; then export EDGL_LIB_PATH=$GRAFT_REPO_ROOT/variants/lib_allsb.so; else unset EDGL_LIB_PATH; fi
python bench.py --workload recipe --steps 100 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L recipe', d['ms_per_step'])"
done; done
EDGL_LIB_PATH=$GRAFT_REPO_ROOT/variants/lib_allsb.so python -m pytest tests/test_gpu_ops.py -x -q -k gemm 2>&1 | tail -1

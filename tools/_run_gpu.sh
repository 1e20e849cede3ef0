#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
for L in new old; do
if [ $L = old ]; then export EDGL_LIB_PATH=$GRAFT_REPO_ROOT/variants/lib_oldgemm.so; else unset EDGL_LIB_PATH; fi
python bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', d['ms_per_step'], d['step_ms_hipevents']['median'])"
done; done

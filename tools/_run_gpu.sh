#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do for E in 0 1 2; do
echo -n "EARLY=$E "; EDGL_LABEL_EARLY=$E python bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['step_ms_hipevents']['median'])"
done; done

#!/bin/bash
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
for v in "0 0" "1 0" "0 1"; do
  set -- $v
  EDGL_ENGINE_LEGACY_FORK=$1 EDGL_DROPBITS=$2 python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('legacy=$1 dropbits=$2', j['ms_per_step'], j['step_ms_hipevents']['median'], j['roofline_attention']['forward']['avg_ms'], j['roofline_attention']['backward']['avg_ms'])"
done
done

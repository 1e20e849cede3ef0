#!/bin/bash
# On the GPU box: A/B of one environment switch inside the real step, alternating, N repeats each.
#   bash tools/ab_env.sh VAR A B [reps] [extra bench args]   ->  ms_per_step, attention forward / backward, dominant kernel per run
VAR=$1; A=$2; B=$3; REPS=${4:-3}; shift 4
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for i in $(seq 1 $REPS); do
  for v in "$A" "$B"; do
    env $VAR=$v python $ROOT/bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); a=d.get('roofline_attention',{})
        print('$VAR=$v', 'ms_per_step', d['ms_per_step'], 'median', d['step_ms_hipevents']['median'], 'att_fwd', a.get('forward',{}).get('avg_ms'), 'att_bwd', a.get('backward',{}).get('avg_ms'), 'dom', d['roofline']['avg_launch_ms'], 'loss', d['loss'])
"
  done
done

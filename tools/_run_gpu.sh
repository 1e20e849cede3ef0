#!/bin/bash
cd $GRAFT_REPO_ROOT
EDGL_BENCH_DUMP_GROUPS=1 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras 2>&1 | grep "step groups\|metric" | cut -c1-260
EDGL_BENCH_DUMP_GROUPS=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>&1 | grep "step groups\|metric" | cut -c1-260
EDGL_BENCH_DUMP_GROUPS=1 python bench.py --no-cpu-baseline --no-extras 2>&1 | grep "step groups\|metric" | cut -c1-260

#!/bin/bash
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py -x -q -m gpu -k "gemm or engine" 2>&1 | tail -2
KT_LINES=40 bash tools/ktrace.sh --workload recipe > gpurun_out/tnp_recipe.txt 2>&1
grep -E "tn_gemm_kernel|reduce_rows" gpurun_out/tnp_recipe.txt | cut -c1-50,55-140; grep -o '"ms_per_step": [0-9.]*' gpurun_out/tnp_recipe.txt | head -1
KT_LINES=40 bash tools/ktrace.sh > gpurun_out/tnp_head.txt 2>&1
grep -E "tn_gemm_group|reduce_rows" gpurun_out/tnp_head.txt | cut -c1-50,55-140; grep -o '"ms_per_step": [0-9.]*' gpurun_out/tnp_head.txt | head -1
python bench.py --workload recipe --steps 100 --warmup 20 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | cut -c1-400

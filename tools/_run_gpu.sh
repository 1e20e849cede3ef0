cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3a
timeout 300 python tools/_dbg_strip.py > gpurun_out/r3a/dbg.txt 2>&1
grep -c nan gpurun_out/r3a/dbg.txt; grep -c "e+[0-9]" gpurun_out/r3a/dbg.txt
timeout 600 python -m pytest tests/test_gpu_score_strip.py -x -q 2>&1 | tail -15
timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -k "score" 2>&1 | tail -5
timeout 120 python tools/strip_bench.py 2>&1 | tail -1
ZERO=0 timeout 120 python tools/strip_bench.py 2>&1 | tail -1

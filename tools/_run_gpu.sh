cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_score_strip.py -x -q 2>&1 | tail -3
EDGL_LIB_PATH=tools/variants/lib_timing.so timeout 120 python tools/strip_probe.py 2>&1 | grep "wave 0\|per MFMA\|whole"
timeout 120 python tools/strip_bench.py 2>&1 | tail -1

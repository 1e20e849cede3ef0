cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_distributed.py -x -q 2>&1 | grep -i "assert\|dict\|passed\|failed" | head -10

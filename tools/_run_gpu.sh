cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_headline_parity.py tests/test_gpu_engine.py tests/test_gpu_model.py -q -m gpu 2>&1 | tail -40

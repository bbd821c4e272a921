cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k flash_attn 2>&1 | tail -2

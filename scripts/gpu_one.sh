cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity_onehot.py -m gpu -q 2>&1 | grep -E "Error|error|FAILED|passed|failed" | head -20

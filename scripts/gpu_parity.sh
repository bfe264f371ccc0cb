set -x
cd $GRAFT_REPO_ROOT
python -c "import re2nn_seq_amd._lib as L; lib=L.load(); print('devices', lib.farnn_device_count())"
timeout 900 python -m pytest tests/test_gpu_parity_onehot.py -m gpu -x -q 2>&1 | tail -40

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03p
FARNN_SHAPE_SEED=${SEED:-11} FARNN_SHAPE_SOAK=${N:-300} timeout 900 python -m pytest tests/test_gpu_chain_viterbi.py -m gpu -q -x --timeout=800 -p no:cacheprovider > gpurun_out/r03p/cv.log 2>&1; echo "rc=$?"
tail -30 gpurun_out/r03p/cv.log | cut -c1-300

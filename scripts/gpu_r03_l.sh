cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03l
FARNN_SHAPE_SEED=${SEED:-7} FARNN_SHAPE_SOAK=${N:-4000} timeout 900 python -m pytest tests/test_gpu_chain_regs_shapes.py -m gpu -q -x --timeout=800 -p no:cacheprovider > gpurun_out/r03l/shapes.log 2>&1; echo "rc=$?"
tail -30 gpurun_out/r03l/shapes.log | cut -c1-300

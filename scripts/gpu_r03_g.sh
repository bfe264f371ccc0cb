cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03g
timeout 900 python -m pytest tests/test_gpu_handoff_soak.py -m gpu -q --timeout=240 -p no:cacheprovider --durations=10 > gpurun_out/r03g/soak.log 2>&1; echo "rc=$?"
tail -25 gpurun_out/r03g/soak.log

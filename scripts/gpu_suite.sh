# the whole GPU suite (no -x: every failure is listed); per-test timeout so that a hung kernel costs minutes, not the budget
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/suite
timeout 2400 python -m pytest tests -m gpu -q --timeout=180 -p no:cacheprovider > gpurun_out/suite/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/suite/pytest.log
tail -25 gpurun_out/suite/pytest.log

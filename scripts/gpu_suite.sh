# the whole GPU suite (no -x: every failure is listed)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/suite
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/suite/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/suite/pytest.log
tail -15 gpurun_out/suite/pytest.log

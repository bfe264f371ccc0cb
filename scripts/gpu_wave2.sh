cd $GRAFT_REPO_ROOT
FARNN_DBG=4096 python bench.py --workload decomp --steps 3 --warmup 1 --no-cpu-baseline --no-pipelined --no-parity 2>&1 | grep "regs kernel" | head -8

cd $GRAFT_REPO_ROOT
FARNN_DBG=16384 python bench.py --workload decomp --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs --no-parity 2>&1 | grep "score tile" | head -8
FARNN_DBG=16384 python bench.py --workload ifst --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs --no-parity 2>&1 | grep "score tile" | head -8

cd $GRAFT_REPO_ROOT
run() { env "$@" python bench.py --workload ifst --steps 500 --warmup 30 --no-cpu-baseline --no-other-configs --no-parity 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*: step %.1f us  kernel %.1f  2-stream %.1f' % (d['ms_per_step']*1e3, d['roofline']['chain_avg_us'], d['pipelined']['ms_per_step']*1e3))"; }
run FARNN_DBG=0
run FARNN_DBG=128
run FARNN_FUSE_SPIN=0
run FARNN_FUSE_SPIN=40
run FARNN_FUSE_SPIN=1000
run FARNN_DBG=2048
run FARNN_DBG=32
run FARNN_DBG=64

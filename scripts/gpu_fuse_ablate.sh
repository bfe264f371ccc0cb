cd $GRAFT_REPO_ROOT
run() { env "$@" python bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-other-configs --no-parity --no-pipelined 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-40s step %.1f us  chain %.1f' % (sys.argv[1], d['ms_per_step']*1e3, d['roofline']['chain_avg_us']))" "$*"; }
run FARNN_DBG=0
run FARNN_DBG=128
run FARNN_DBG=32
run FARNN_DBG=64
run FARNN_DBG=96
run FARNN_DBG=112
run FARNN_DBG=0 FARNN_FUSE_SPIN=1000

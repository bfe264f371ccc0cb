cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu-baseline --no-other-configs --no-parity --no-pipelined "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-44s step %.1f us  kernel %.1f  timed %d' % (sys.argv[1], d['ms_per_step']*1e3, d['roofline']['chain_avg_us'], d['roofline']['launches_timed']))" "$*"; }
run --steps 20 --warmup 5 --event-stride 0
run --steps 20 --warmup 5 --event-stride 2
run --steps 20 --warmup 5 --event-stride 1
run --steps 20 --warmup 50 --event-stride 2
run --steps 20 --warmup 500 --event-stride 2
run --steps 200 --warmup 20 --event-stride 2
run --steps 200 --warmup 20 --event-stride 0

cd $GRAFT_REPO_ROOT
run() { env "$@" python bench.py --workload ifst --steps 500 --warmup 30 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*: step %.1f us  kernel %.1f + %.1f 2-stream %.1f parity %s' % (d['ms_per_step']*1e3, d['roofline']['chain_avg_us'], d['roofline']['score_decode_avg_us'], d['pipelined']['ms_per_step']*1e3, d['parity']['tags_equal']))"; }
run FARNN_XCD_PAIR=1
run FARNN_XCD_PAIR=0
run FARNN_XCD_PAIR=1 FARNN_NOFUSE=1
run FARNN_XCD_PAIR=0 FARNN_NOFUSE=1
run FARNN_XCD_PAIR=1 FARNN_DBG=128
FARNN_DBG=16384 python bench.py --workload ifst --steps 1 --warmup 1 --no-cpu-baseline --no-other-configs --no-parity 2>&1 | grep "score tile" | head -4

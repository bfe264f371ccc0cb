cd $GRAFT_REPO_ROOT
for dbg in 0 128; do
FARNN_DBG=$dbg python bench.py --workload decomp --steps 500 --warmup 30 --no-cpu-baseline --no-other-configs --no-parity 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('DBG=$dbg step %.1f us  recurrence %.1f  score %.1f  2-stream %.1f %.3e tok/s' % (d['ms_per_step']*1e3, d['roofline']['chain_avg_us'], d['roofline']['score_decode_avg_us'], d['pipelined']['ms_per_step']*1e3, d['value']))"
done
FARNN_DBG=4096 python bench.py --workload decomp --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs --no-parity 2>&1 | grep "regs kernel" | head -8

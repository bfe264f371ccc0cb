cd $GRAFT_REPO_ROOT
for es in -1 0 4 1; do for rep in 1 2; do
python bench.py --steps 20 --warmup 5 --event-stride $es --no-cpu-baseline --no-other-configs --no-parity 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('stride $es: step %.1f us  kernel %.1f timed %d  2-stream %.1f' % (d['ms_per_step']*1e3, d['roofline']['chain_avg_us'], d['roofline']['launches_timed'], d['pipelined']['ms_per_step']*1e3))"
done; done
python bench.py --steps 200 --warmup 5 --event-stride 0 --no-cpu-baseline --no-other-configs --no-parity 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('200 steps stride 0: step %.1f us' % (d['ms_per_step']*1e3))"

cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity_decomposed.py tests/test_gpu_parity_bench_size.py -q -k "decomposed and not independent" 2>&1 | tail -3
for nf in 0 1; do
FARNN_NOFUSE=$nf python bench.py --workload decomp --steps 500 --warmup 30 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('NOFUSE=$nf step %.1f us  recurrence %.1f  score %.1f  2-stream %.1f parity %s  %.3e tok/s' % (d['ms_per_step']*1e3, d['roofline']['chain_avg_us'], d['roofline']['score_decode_avg_us'], d['pipelined']['ms_per_step']*1e3, d['parity']['tags_equal'], d['value']))"
done

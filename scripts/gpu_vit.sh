cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity_decomposed.py tests/test_gpu_parity_bench_size.py -q -k "viterbi or crf" 2>&1 | tail -3
python bench.py --workload ifst_crf --steps 300 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('step %.1f us  chain %.1f  score %.1f  2-stream %.1f parity %s' % (d['ms_per_step']*1e3, d['roofline']['chain_avg_us'], d['roofline']['score_decode_avg_us'], d['pipelined']['ms_per_step']*1e3, d['parity']['tags_equal']))"

cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity_decomposed.py tests/test_gpu_parity_bench_size.py -x -q 2>&1 | tail -5
run() { env "$@" python bench.py --workload decomp --steps 300 --warmup 20 --no-cpu-baseline --no-pipelined 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-40s step %.1f us  %s chain %.1f  score %.1f  parity %s err %s' % (sys.argv[1], d['ms_per_step']*1e3, d['roofline']['kernel'], d['roofline']['chain_avg_us'], d['roofline']['score_decode_avg_us'], d['parity']['tags_equal'], d['parity']['max_score_err']))" "$*"; }
run FARNN_DECOMP_NOREGS=1
run FARNN_X=0



cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity_bench_size.py tests/test_gpu_parity_decomposed.py tests/test_gpu_cli_e2e.py -q 2>&1 | tail -3
run() { env $1 python bench.py --workload decomp --rank $2 --farnn $3 --batch $4 --steps 60 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*: step %.1f us  kernel %.1f + %.1f 2-stream %.1f parity %s %.3e' % (d['ms_per_step']*1e3, d['roofline']['chain_avg_us'], d['roofline']['score_decode_avg_us'], d['pipelined']['ms_per_step']*1e3, d['parity']['tags_equal'], d['value']))"; }
run X=0 250 2 256
run X=0 150 2 256
run FARNN_ROWS_NOREGS=1 150 2 256
run X=0 100 1 256
run X=0 100 0 256
run FARNN_ROWS_NOREGS=1 100 0 256
run X=0 50 2 256
run FARNN_ROWS_NOREGS=1 50 2 256

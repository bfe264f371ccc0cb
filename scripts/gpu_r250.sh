cd $GRAFT_REPO_ROOT
run() { env "$@" python bench.py --workload decomp --rank 250 --farnn 2 --steps 60 --warmup 5 --no-cpu-baseline --no-other-configs --no-parity 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*: step %.1f us  kernel %.1f + %.1f 2-stream %.1f' % (d['ms_per_step']*1e3, d['roofline']['chain_avg_us'], d['roofline']['score_decode_avg_us'], d['pipelined']['ms_per_step']*1e3))"; }
run FARNN_ROWS_NSEQ=0
run FARNN_ROWS_NSEQ=1
run FARNN_ROWS_NSEQ=2
run FARNN_ROWS_NSEQ=4

cd $GRAFT_REPO_ROOT
run() { env $1 python bench.py --workload decomp --rank $2 --farnn $3 --steps 60 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*: step %.1f us  kernel %.1f + %.1f 2-stream %.1f parity %s %.3e' % (d['ms_per_step']*1e3, d['roofline']['chain_avg_us'], d['roofline']['score_decode_avg_us'], d['pipelined']['ms_per_step']*1e3, d['parity']['tags_equal'], d['value']))"; }
run FARNN_ROWS_NSEQ=2 250 2
run FARNN_ROWS_NSEQ=2 100 1
run FARNN_ROWS_NSEQ=2 100 2

cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity_decomposed.py tests/test_gpu_cli_e2e.py -m gpu -x -q 2>&1 | tail -12
run() {
  timeout 300 python bench.py --workload decomp --rank $1 --farnn $2 --steps 200 --warmup 10 --no-pipelined --event-stride 4 $4 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$3 R=$1 farnn=$2 $4', 'tok/s %.3e' % d['value'], 'ms/step %.4f' % d['ms_per_step'], 'chain_us %.1f score_us %.1f' % (r['chain_avg_us'], r['score_decode_avg_us']))"
}
for cfg in "50 0" "50 2" "100 1" "250 0" "250 2"; do
  set -- $cfg
  run $1 $2 new ""
done
FARNN_ROWS_NSEQ=2 run 50 0 nseq2 ""
FARNN_ROWS_NSEQ=2 run 250 2 nseq2 ""

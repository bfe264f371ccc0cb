cd $GRAFT_REPO_ROOT
run() {
  timeout 300 python bench.py --steps 300 --warmup 20 --no-pipelined --no-cpu-baseline --event-stride 4 $2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1 $2', 'tok/s %.3e' % d['value'], 'ms/step %.4f' % d['ms_per_step'], 'chain_us %.1f score_us %.1f' % (r['chain_avg_us'], r['score_decode_avg_us']))"
}
for nld in 3 4 5; do
  FARNN_NLD=$nld run "nld$nld" ""
  FARNN_NLD=$nld run "nld$nld" "--full-length"
done
FARNN_NLD=5 FARNN_KS=4 run "nld5 ks4" ""
FARNN_NLD=5 timeout 600 python -m pytest tests/test_gpu_parity_onehot.py -m gpu -x -q 2>&1 | tail -2

cd $GRAFT_REPO_ROOT
run() {
  timeout 300 python bench.py --steps 300 --warmup 20 --no-pipelined --no-cpu-baseline --event-stride 4 $2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1 $2', 'ms/step %.4f' % d['ms_per_step'], 'chain_us %.1f score_us %.1f' % (r['chain_avg_us'], r['score_decode_avg_us']))"
}
for dbg in 0 16 32 48 64 112; do
  FARNN_DBG=$dbg run "dbg$dbg" ""
done
FARNN_DBG=0 run "full" "--full-length"
FARNN_DBG=112 run "full dbg112" "--full-length"

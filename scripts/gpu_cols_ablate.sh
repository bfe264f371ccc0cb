cd $GRAFT_REPO_ROOT
run() {
  timeout 300 python bench.py --steps 200 --warmup 20 --no-pipelined --no-cpu-baseline --event-stride 4 --full-length 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', 'chain_us %.1f' % r['chain_avg_us'])"
}
for dbg in 0 1 2 4 8 3 5 7 15; do
  FARNN_DBG=$dbg run "dbg$dbg"
done

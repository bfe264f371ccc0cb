cd $GRAFT_REPO_ROOT
run() {
  timeout 300 python bench.py --workload decomp --rank $1 --farnn $2 --steps $3 --warmup 10 --no-pipelined 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$4 R=$1 farnn=$2', 'tok/s %.3e' % d['value'], 'ms/step %.4f' % d['ms_per_step'])"
}
for dbg in 0 1 2 4 8 3 7 15; do
  export FARNN_DBG=$dbg
  run 50 0 200 dbg$dbg
done
for n in 1 2 4; do
  export FARNN_DBG=0 FARNN_ROWS_NSEQ=$n
  run 50 0 200 nseq$n
done

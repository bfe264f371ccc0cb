cd $GRAFT_REPO_ROOT
run() {
  timeout 300 python bench.py --workload decomp --rank $1 --farnn $2 --steps 200 --warmup 10 --no-pipelined --event-stride 4 $4 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('$3 R=$1 farnn=$2 $4', 'ms/step %.4f' % d['ms_per_step'], 'chain_us %.1f score_us %.1f' % (r['chain_avg_us'], r['score_decode_avg_us']))"
}
for L in 8 16 32 64 128; do
  run 50 0 full "--full-length --seqlen $L"
done
for L in 16 64; do
  FARNN_DBG=15 run 50 0 dbg15 "--full-length --seqlen $L"
  FARNN_DBG=1 run 50 0 dbg1 "--full-length --seqlen $L"
done
for L in 16 64; do
  run 250 2 full "--full-length --seqlen $L"
done

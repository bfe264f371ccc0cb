cd $GRAFT_REPO_ROOT
for dbg in 0 1 2 3 4 8 9 11 7; do
  echo "DBG=$dbg batch=64 full"
  FARNN_DBG=$dbg python bench.py --steps 100 --warmup 10 --no-cpu-baseline --full-length --batch 64 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('chain_us %.1f score_us %.1f' % (r['kernel_avg_us'], r['score_decode_avg_us']))"
done

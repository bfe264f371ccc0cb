cd $GRAFT_REPO_ROOT
pr() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1 chain_us %.1f score_us %.1f ms/step %.4f' % (r['kernel_avg_us'], r['score_decode_avg_us'], d['ms_per_step']))"; }
for dbg in 0 1 2 3 8 9 10 11 4; do
  echo "DBG=$dbg"
  FARNN_DBG=$dbg python bench.py --steps 100 --warmup 10 --no-cpu-baseline --event-stride 1 --full-length --batch 64 2>/dev/null | pr b64
  FARNN_DBG=$dbg python bench.py --steps 100 --warmup 10 --no-cpu-baseline --event-stride 1 2>/dev/null | pr ragged
done
for L in 8 16 32; do
  python bench.py --steps 100 --warmup 10 --no-cpu-baseline --event-stride 1 --full-length --batch 64 --seqlen $L 2>/dev/null | pr "b64_L$L"
done
python bench.py --steps 200 --warmup 10 --no-cpu-baseline 2>/dev/null | pr "ragged_stride8"

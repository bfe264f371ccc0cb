cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity_onehot.py -m gpu -x -q 2>&1 | tail -3
for L in 8 16 32 64 128; do
  echo "seqlen=$L full-length"
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline --full-length --seqlen $L 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('tok/s %.3g ms/step %.4f chain_us %.1f frac %.3f score_us %.1f' % (d['value'], d['ms_per_step'], r['chain_avg_us'], r['frac'], r['score_decode_avg_us']))"
done
for B in 64 128 512 1024; do
  echo "batch=$B seqlen=64 full-length"
  python bench.py --steps 100 --warmup 10 --no-cpu-baseline --full-length --batch $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('tok/s %.3g ms/step %.4f chain_us %.1f frac %.3f score_us %.1f' % (d['value'], d['ms_per_step'], r['chain_avg_us'], r['frac'], r['score_decode_avg_us']))"
done

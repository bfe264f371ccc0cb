cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity_onehot.py -m gpu -x -q 2>&1 | tail -5
pr() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1 tok/s %.3g ms/step %.4f chain_us %.1f frac %.3f score_us %.1f' % (d['value'], d['ms_per_step'], r['chain_avg_us'], r['frac'], r['score_decode_avg_us']))"; }
for nld in 1 2; do for ks in 2 3 4; do for rpg in 4 8 12; do
  echo "NLD=$nld KS=$ks RPG=$rpg"
  FARNN_NLD=$nld FARNN_KS=$ks FARNN_RPG=$rpg python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | pr ragged
  FARNN_NLD=$nld FARNN_KS=$ks FARNN_RPG=$rpg python bench.py --steps 100 --warmup 10 --no-cpu-baseline --full-length 2>/dev/null | pr full
  FARNN_NLD=$nld FARNN_KS=$ks FARNN_RPG=$rpg python bench.py --steps 100 --warmup 10 --no-cpu-baseline --full-length --batch 64 2>/dev/null | pr b64
done; done; done 2>&1 | tee gpurun_out/sweep2.txt

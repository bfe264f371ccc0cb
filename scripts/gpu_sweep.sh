set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
python __graft_entry__.py smoke 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_parity_onehot.py -m gpu -x -q 2>&1 | tail -5
for pf in 2 4 6 8; do for rpg in 4 8 12; do
  echo "PF=$pf RPG=$rpg"
  FARNN_PF=$pf FARNN_RPG=$rpg python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('ragged tok/s %.3g ms/step %.4f chain_us %.1f frac %.3f score_us %.1f' % (d['value'], d['ms_per_step'], r['chain_avg_us'], r['frac'], r['score_decode_avg_us']))"
  FARNN_PF=$pf FARNN_RPG=$rpg python bench.py --steps 200 --warmup 20 --no-cpu-baseline --full-length 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('full   tok/s %.3g ms/step %.4f chain_us %.1f frac %.3f score_us %.1f' % (d['value'], d['ms_per_step'], r['chain_avg_us'], r['frac'], r['score_decode_avg_us']))"
done; done 2>&1 | tee gpurun_out/sweep1.txt

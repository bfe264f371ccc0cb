cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity_onehot.py -m gpu -x -q 2>&1 | tail -3
pr() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1 tok/s %.3g ms/step %.4f chain_us %.1f frac %.3f score_us %.1f' % (d['value'], d['ms_per_step'], r['kernel_avg_us'], r['frac'], r['score_decode_avg_us']))"; }
for cfg in "FARNN_NLD=2" "FARNN_NLD=3" "FARNN_NLD=4" "FARNN_NLD=3 FARNN_RPG=12" "FARNN_NLD=4 FARNN_RPG=6"; do
echo "== $cfg"
env $cfg python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | pr ragged
env $cfg python bench.py --steps 100 --warmup 10 --no-cpu-baseline --full-length 2>/dev/null | pr full
env $cfg python bench.py --steps 100 --warmup 10 --no-cpu-baseline --full-length --batch 64 2>/dev/null | pr b64
done

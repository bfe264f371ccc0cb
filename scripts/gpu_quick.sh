cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
pr() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1 tok/s %.3g ms/step %.4f chain_us %.1f frac %.3f score_us %.1f' % (d['value'], d['ms_per_step'], r['kernel_avg_us'], r['frac'], r['score_decode_avg_us']))"; }
for cfg in "FARNN_NOSORT=0" "FARNN_NOSORT=1"; do
echo "== $cfg"
env $cfg python bench.py --steps 200 --warmup 10 --no-cpu-baseline 2>/dev/null | pr ragged
env $cfg python bench.py --steps 200 --warmup 10 --no-cpu-baseline --full-length 2>/dev/null | pr full
env $cfg python bench.py --steps 200 --warmup 10 --no-cpu-baseline --batch 1024 2>/dev/null | pr ragged_b1024
done
python bench.py 2>/dev/null

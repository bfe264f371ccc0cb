cd $GRAFT_REPO_ROOT
pr() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1 tok/s %.3g ms/step %.4f chain_us %.1f frac %.3f score_us %.1f' % (d['value'], d['ms_per_step'], r['chain_avg_us'], r['frac'], r['score_decode_avg_us']))"; }
for st in 1 2 3 1 2; do
python bench.py --steps 300 --warmup 20 --no-cpu-baseline --streams $st 2>/dev/null | pr "streams=$st"
done
python bench.py --steps 300 --warmup 20 --no-cpu-baseline --streams 2 --full-length 2>/dev/null | pr "full streams=2"
python bench.py --steps 300 --warmup 20 --no-cpu-baseline --streams 1 --full-length 2>/dev/null | pr "full streams=1"

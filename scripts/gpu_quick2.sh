cd $GRAFT_REPO_ROOT
pr() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1 tok/s %.3g ms/step %.4f chain_us %.1f frac %.3f score_us %.1f' % (d['value'], d['ms_per_step'], r['kernel_avg_us'], r['frac'], r['score_decode_avg_us']))"; }
for o in "" sorted fold; do
for rep in 1 2; do
FARNN_BENCH_ORDER=$o python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | pr "order=$o"
done; done

cd $GRAFT_REPO_ROOT
pr() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1 tok/s %.3g ms/step %.4f chain_us %.1f score_us %.1f' % (d['value'], d['ms_per_step'], r['kernel_avg_us'], r['score_decode_avg_us']))"; }
python bench.py --steps 300 --warmup 20 --no-cpu-baseline 2>/dev/null | pr events
FARNN_BENCH_NOEVENTS=1 python bench.py --steps 300 --warmup 20 --no-cpu-baseline 2>/dev/null | pr noevents
python bench.py --steps 300 --warmup 20 --no-cpu-baseline 2>/dev/null | pr events
FARNN_BENCH_NOEVENTS=1 python bench.py --steps 300 --warmup 20 --no-cpu-baseline 2>/dev/null | pr noevents

cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
pr() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1 tok/s %.3g ms/step %.4f chain_us %.1f frac %.3f score_us %.1f' % (d['value'], d['ms_per_step'], r['chain_avg_us'], r['frac'], r['score_decode_avg_us']), (d.get('pipelined') or {}).get('value'))"; }
python bench.py --steps 50 --warmup 5 --no-cpu-baseline --workload decomp 2>/dev/null | pr decomp_R50
python bench.py --steps 50 --warmup 5 --no-cpu-baseline --workload decomp --rank 100 2>/dev/null | pr decomp_R100
FARNN_DECOMP_GENERIC=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload decomp 2>/dev/null | pr decomp_R50_generic
python bench.py --steps 50 --warmup 5 --no-cpu-baseline --workload decomp --rank 200 2>/dev/null | pr decomp_R200

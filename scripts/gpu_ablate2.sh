cd $GRAFT_REPO_ROOT
pr() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1 chain_us %.1f score_us %.1f ms/step %.4f' % (r['kernel_avg_us'], r['score_decode_avg_us'], d['ms_per_step']))"; }
for dbg in 0 16 32 48 64 112; do
  FARNN_DBG=$dbg python bench.py --steps 100 --warmup 10 --no-cpu-baseline --event-stride 1 2>/dev/null | pr "DBG=$dbg ragged"
done
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/k2prof -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --event-stride 0 > /dev/null 2>&1
head -5 gpurun_out/k2prof/*/*kernel_stats.csv

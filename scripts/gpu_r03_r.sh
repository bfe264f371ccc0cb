# mixed register forms of the rows kernel: parity at bench size + timing of the S = 134 shapes (and the FARNN_ROWS_NOREGS / form-4 baselines)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03r; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity_bench_size.py -m gpu -q -x --timeout=300 -p no:cacheprovider -k "decomposed_ifst_at_bench_size" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log | cut -c1-300
Q="--steps 60 --warmup 10 --no-cpu-baseline --no-pipelined --no-other-configs"
for cfg in "150 134" "250 134" "250 104" "150 104"; do
  set -- $cfg
  timeout 300 python bench.py --workload decomp --rank $1 --farnn 2 --states $2 $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('R=$1 S=$2', '%.3e' % d['value'], '%.1f us' % (d['ms_per_step']*1e3), d['roofline']['kernel'], '%.1f' % d['roofline']['kernel_avg_us'], d['parity']['tags_equal'], d['parity'].get('max_score_err'))"
done

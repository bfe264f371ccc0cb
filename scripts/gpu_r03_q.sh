cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity_onehot.py tests/test_gpu_chain_regs_shapes.py tests/test_gpu_chain_viterbi.py -m gpu -q -x --timeout=300 -p no:cacheprovider 2>&1 | tail -3
Q="--steps 300 --warmup 30 --no-cpu-baseline --no-pipelined --no-other-configs"
for rep in 1 2 3; do
timeout 200 python bench.py $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ragged headline', '%.4e' % d['value'], '%.2f us' % (d['ms_per_step']*1e3), '%.2f' % d['roofline']['kernel_avg_us'], d['parity']['tags_equal'])"
done
timeout 200 python bench.py $Q --full-length 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('full-length', '%.4e' % d['value'], '%.2f us' % (d['ms_per_step']*1e3), d['parity']['tags_equal'])"
FARNN_NOFUSE=1 timeout 200 python bench.py $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('two-kernel', '%.4e' % d['value'], '%.2f us' % (d['ms_per_step']*1e3), 'chain %.2f' % d['roofline']['chain_avg_us'])"
timeout 200 python bench.py --workload ifst_crf $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ifst_crf', '%.4e' % d['value'], '%.2f us' % (d['ms_per_step']*1e3), d['parity']['tags_equal'])"

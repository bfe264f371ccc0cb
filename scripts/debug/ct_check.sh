# the compact one-launch kernel: its parity tests (twice) and the bench's compact line at three shapes
cd $GRAFT_REPO_ROOT
for i in 1 2; do timeout 300 python -m pytest tests/test_gpu_parity_onehot.py -q -k "compact" 2>&1 | grep -E "passed|failed"; done > gpurun_out/ct_test.txt
Q="--no-cpu-baseline --no-other-configs --no-pipelined"
for a in "" "--full-length" "--batch 1024"; do
timeout 120 python bench.py $a $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['compact']; print('dense %.2f us; compact %.2f us per step, kernel %.2f us, tags equal %s' % (d['ms_per_step']*1e3, c['ms_per_step']*1e3, c['kernel_avg_us'], c['tags_equal_dense']))"
done > gpurun_out/ct_bench.txt

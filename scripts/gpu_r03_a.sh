# round 3, first contact of the register-fed recurrence kernel: smoke, onehot parity, A/B bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03a
O=gpurun_out/r03a
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
tail -3 $O/smoke.log
if ! grep -q "smoke ok" $O/smoke.log; then exit 1; fi
timeout 900 python -m pytest tests/test_gpu_parity_onehot.py -m gpu -q -x > $O/onehot.log 2>&1; echo "rc=$?" >> $O/onehot.log
tail -15 $O/onehot.log
timeout 300 python bench.py --steps 200 --warmup 20 > $O/bench_regs.json 2> $O/bench_regs.err; echo "rc=$?"
FARNN_NOREGS=1 timeout 300 python bench.py --steps 200 --warmup 20 > $O/bench_r02.json 2> $O/bench_r02.err; echo "rc=$?"
python - <<'PY'
import json
for n in ('regs','r02'):
    try:
        d=json.loads(open(f'gpurun_out/r03a/bench_{n}.json').read().strip().splitlines()[-1])
        print(n, d['value'], d['ms_per_step'], d['roofline'].get('kernel'), d['roofline'].get('kernel_avg_us'), d.get('parity'))
    except Exception as e:
        print(n, 'failed', e)
PY

# round 3: whole GPU suite + the headline bench in the driver's form + kernel stats
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03h
O=gpurun_out/r03h
timeout 1500 python -m pytest tests -m gpu -q --timeout=240 -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.log)"
grep -E "^FAILED|^ERROR" $O/pytest.log | head
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench_driver_form.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03h/bench_driver_form.json').read().splitlines() if l.startswith('{')][-1])
r=d['roofline']
print('value %.4e ms/step %.4f kernel %s avg %.2f us frac %.3f bound %s' % (d['value'], d['ms_per_step'], r['kernel'], r['kernel_avg_us'], r['frac'], r['bound']))
print('parity', d['parity'])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
for o in d['other_configs']:
    print(' ', o['workload'], '%.3e' % o['value'], o.get('ms_per_step'), o['roofline']['kernel'], o.get('parity'))
print('pipelined', d.get('pipelined'))
PY

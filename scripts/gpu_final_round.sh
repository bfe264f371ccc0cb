# End-of-round check on the GPU box: what the driver runs (GPU suite, smoke(), the default bench line in its own form, timed).
cd $GRAFT_REPO_ROOT
O=gpurun_out/final
rm -rf $O; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
s0=$(date +%s.%N)
timeout 1200 python bench.py --steps 20 --warmup 5 2>$O/driver.err | tail -1 > $O/driver.json
s1=$(date +%s.%N)
python - "$s0" "$s1" <<'PY'
import json, sys
d = json.load(open('gpurun_out/final/driver.json')); r = d['roofline']
print('driver form: %.3e tok/s, %.2f us per step, dominant kernel %s %.1f us, frac %.3f (all-L2 %.3f)%s; wall %.0f s' % (
    d['value'], d['ms_per_step'] * 1e3, r['kernel'], r['kernel_avg_us'], r['frac'], r.get('frac_all_l2', 0),
    ' model_falsified' if r.get('model_falsified') else '', float(sys.argv[2]) - float(sys.argv[1])))
c = d['cpu_baseline']
print('cpu_baseline: %.3e tok/s on %d of %d threads; by threads %s; two-phase form best %.3e' % (
    c['value'], c['cores'], c['host_threads'], c['rate_by_threads'], c['two_phase_form_best']))
print('compact:', {k: (round(v, 3) if isinstance(v, float) else v) for k, v in d.get('compact', {}).items() if k != 'note'})
for o in d['other_configs']:
    rr = o.get('roofline', {})
    print('  %-44s %.3e tok/s %8.2f us  parity %s  wall %.1f s %s' % (o['workload'], o.get('value', 0), o.get('ms_per_step', 0) * 1e3,
          o.get('parity', {}).get('tags_equal'), o.get('wall_s', 0), o.get('error', '')))
PY

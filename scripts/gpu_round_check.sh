# Full GPU suite + the bench lines that go under profiles/ (run on the GPU box through gpurun).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/bench
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/bench/pytest_gpu.txt
cat gpurun_out/bench/pytest_gpu.txt
for wl in decomp1 decomp0; do
  python bench.py --workload $wl --steps 100 --warmup 10 2>gpurun_out/bench/$wl.err | tail -1 > gpurun_out/bench/$wl.json
done
python bench.py --steps 200 --warmup 20 2>gpurun_out/bench/ifst.err | tail -1 > gpurun_out/bench/ifst.json
python - <<'PY'
import json
for wl in ('ifst', 'decomp1', 'decomp0'):
    d = json.load(open('gpurun_out/bench/%s.json' % wl)); r = d['roofline']
    print(wl, '%.3e' % d['value'], 'ms/step %.3f' % d['ms_per_step'], r['kernel'], r['bound'], 'achieved %.1f %s frac %.3f' % (r['achieved'], r['unit'], r['frac']),
          'chain %.1f score %.1f' % (r['chain_avg_us'], r['score_decode_avg_us']), 'pipelined %.3e' % d.get('pipelined', {}).get('value', 0))
PY

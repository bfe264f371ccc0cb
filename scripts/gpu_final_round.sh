# End-of-round check on the GPU box: full GPU suite, smoke, the bench lines kept under profiles/.
cd $GRAFT_REPO_ROOT
O=gpurun_out/final
rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 200 --warmup 20 2>$O/ifst.err | tail -1 > $O/ifst.json
python bench.py --workload train --steps 50 --warmup 5 2>$O/train.err | tail -1 > $O/train.json
python bench.py --workload train --crf --steps 50 --warmup 5 --no-cpu-baseline 2>$O/train_crf.err | tail -1 > $O/train_crf.json
python bench.py --workload decomp1 --steps 100 --warmup 10 2>$O/decomp1.err | tail -1 > $O/decomp1.json
python bench.py --workload decomp0 --steps 100 --warmup 10 2>$O/decomp0.err | tail -1 > $O/decomp0.json
python - <<'PY'
import json
for wl in ('ifst', 'train', 'train_crf', 'decomp1', 'decomp0'):
    d = json.load(open('gpurun_out/final/%s.json' % wl)); r = d['roofline']
    print(wl, '%.3e' % d['value'], 'ms/step %.3f' % d['ms_per_step'], r['bound'], 'frac %.3f' % r['frac'], 'kernel %.1f us' % r['kernel_avg_us'])
PY

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03i
O=gpurun_out/r03i
timeout 200 python -m pytest tests/test_gpu_parity_onehot.py tests/test_gpu_handoff_soak.py -m gpu -q --timeout=120 -p no:cacheprovider > $O/t.log 2>&1; echo "rc=$? $(tail -1 $O/t.log)"
for i in 1 2; do timeout 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs --no-pipelined > $O/b$i.json 2>/dev/null; python -c "
import json; d=json.loads([l for l in open('$O/b$i.json').read().splitlines() if l.startswith('{')][-1]); print('%.4e %.4f %s' % (d['value'], d['ms_per_step'], d['parity']['tags_equal']))"; done

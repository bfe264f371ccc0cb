#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04p
rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; tail -5 $O/pytest.log
Q="--steps 300 --warmup 20 --no-other-configs --no-cpu-baseline --no-pipelined"
for i in 1 2; do
python bench.py --workload decomp $Q > $O/decomp_$i.json 2>/dev/null
FARNN_NOLABELMAP=1 python bench.py --workload decomp $Q > $O/decomp_nolm_$i.json 2>/dev/null
done
python bench.py --steps 20 --warmup 5 --cpu-seconds 6 > $O/default_driver.json 2>$O/default_driver.err
python scripts/sumjson.py $O/*.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04p/default_driver.json').read().strip().splitlines()[-1])
print('cpu_baseline', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline']['host_threads'])
print(d['cpu_baseline']['sample'][:200])
print('faithful', d['cpu_baseline_faithful']['value'])
print('host_inclusive', {k:v for k,v in d['host_inclusive'].items() if k!='note'})
PY

# Viterbi / one-launch CRF iteration: CRF parity tests, ifst_crf bench line, phase probe
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03m; mkdir -p $O
timeout 400 python -m pytest tests -m gpu -q -x --timeout=120 -p no:cacheprovider -k "crf or viterbi or Viterbi" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -5 $O/tests.log | cut -c1-300
timeout 200 python bench.py --workload ifst_crf --steps 200 --warmup 20 --no-cpu-baseline --no-pipelined --no-other-configs > $O/bench_crf.json 2>$O/bench_crf.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r03m/bench_crf.json').read().strip().splitlines()[-1])
    print(d['value'], d['ms_per_step'], d.get('parity'))
    print(d.get('roofline'))
except Exception as e:
    print('no bench line', e); print(open('gpurun_out/r03m/bench_crf.err').read()[-1500:])
PY
FARNN_NOFUSE=1 timeout 200 python bench.py --workload ifst_crf --steps 200 --warmup 20 --no-cpu-baseline --no-pipelined --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('two launches:', d['value'], d['ms_per_step'])"

# Viterbi iteration: CRF parity tests, ifst_crf bench line, phase probe
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03m; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q -x --timeout=300 -p no:cacheprovider -k "crf or viterbi or Viterbi" > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
timeout 200 python bench.py --workload ifst_crf --steps 200 --warmup 20 --no-cpu-baseline --no-pipelined --no-other-configs > $O/bench_crf.json 2>$O/bench_crf.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03m/bench_crf.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('parity'))
print({k:v for k,v in d.get('kernels',{}).items()} if 'kernels' in d else d.get('roofline'))
PY
Q="--steps 3 --warmup 2 --no-cpu-baseline --no-pipelined --no-other-configs --no-parity"
FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_probes.so FARNN_DBG=8192 timeout 120 python bench.py --workload ifst_crf $Q 2>/dev/null | grep "^viterbi" | sort | tail -4 | tee $O/probe_viterbi_phases.txt

# round 3: register-fed recurrence -- parity, chain-only timings, in-kernel timeline
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03b
O=gpurun_out/r03b
timeout 150 python -m pytest tests/test_gpu_parity_onehot.py -m gpu -q > $O/onehot.log 2>&1; echo "rc=$?" >> $O/onehot.log
tail -5 $O/onehot.log
run() { n=$1; shift; env "$@" timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-pipelined --no-other-configs > $O/bench_$n.json 2> $O/bench_$n.err; echo "$n rc=$?"; }
run regs FARNN_X=0
run regs2 FARNN_X=1
run regs_nofuse FARNN_NOFUSE=1
python - <<'PY'
import json
for n in ('regs','regs2','regs_nofuse'):
    try:
        d=json.loads(open(f'gpurun_out/r03b/bench_{n}.json').read().strip().splitlines()[-1])
        r=d['roofline']
        print(n, '%.3e'%d['value'], 'ms/step %.4f'%d['ms_per_step'], r.get('kernel'), 'chain %.2f score %.2f'%(r.get('chain_avg_us',0), r.get('score_decode_avg_us',0)), d.get('parity',{}).get('tags_equal'))
    except Exception as e:
        print(n, 'failed', e)
PY
FARNN_DBG=512 FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_probes.so timeout 100 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-pipelined --no-other-configs > $O/probe.json 2> $O/probe.err
grep "^seq" $O/probe.json | tail -16

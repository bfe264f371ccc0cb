cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03j
O=gpurun_out/r03j
timeout 600 python -m pytest tests/test_gpu_parity_decomposed.py tests/test_gpu_parity_bench_size.py -m gpu -q -x --timeout=200 -p no:cacheprovider > $O/t.log 2>&1; echo "rc=$? $(tail -1 $O/t.log)"
grep -E "^E  |^FAILED" $O/t.log | head -12
run() { n=$1; shift; env "$@" timeout 200 python bench.py --workload decomp --steps 300 --warmup 30 --no-cpu-baseline --no-other-configs --no-pipelined > $O/b_$n.json 2>/dev/null; python -c "
import json; d=json.loads([l for l in open('$O/b_$n.json').read().splitlines() if l.startswith('{')][-1]); r=d['roofline']; print('$n %.4e ms/step %.4f %s chain %.1f score %.1f parity %s' % (d['value'], d['ms_per_step'], r['kernel'][:40], r['chain_avg_us'], r['score_decode_avg_us'], d['parity']))"; }
run fused FARNN_X=1
run nofuse FARNN_NOFUSE=1
run four_fused FARNN_DECOMP_FOUR=1
run four_nofuse FARNN_DECOMP_FOUR=1 FARNN_NOFUSE=1

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03k
O=gpurun_out/r03k
run() { n=$1; shift; env "$@" timeout 200 python bench.py --workload decomp --steps 300 --warmup 30 --no-cpu-baseline --no-other-configs --no-pipelined > $O/b_$n.json 2>/dev/null; python -c "
import json; d=json.loads([l for l in open('$O/b_$n.json').read().splitlines() if l.startswith('{')][-1]); r=d['roofline']; print('$n %.4e ms/step %.4f %s chain %.1f score %.1f parity %s' % (d['value'], d['ms_per_step'], r['kernel'][:40], r['chain_avg_us'], r['score_decode_avg_us'], d['parity']['tags_equal']))"; }
run pk_nofuse FARNN_NOFUSE=1
run scalar_nofuse FARNN_NOFUSE=1 FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_abscalar.so
run pk_fused FARNN_X=1
run scalar_fused FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_abscalar.so

# round 3: ablations of the register-fed recurrence (timing only, results wrong by construction)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03c
O=gpurun_out/r03c
rm -f $O/*.json
run() { n=$1; shift; env "$@" timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-pipelined --no-other-configs --no-parity > $O/bench_$n.json 2> $O/bench_$n.err; echo "$n rc=$?"; }
run nofuse_ab0 FARNN_NOFUSE=1
for v in 256 512 95; do run nofuse_ab$v FARNN_NOFUSE=1 FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_ab$v.so; done
run fused_ab0 FARNN_X=1
run fullen_nofuse FARNN_NOFUSE=1 FARNN_BENCH_ARGS=1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03c/bench_*.json')):
    try:
        d=json.loads([l for l in open(f).read().strip().splitlines() if l.startswith('{')][-1])
        r=d['roofline']
        print(f.split('bench_')[1], 'ms/step %.4f'%d['ms_per_step'], 'chain %.2f score %.2f'%(r.get('chain_avg_us',0), r.get('score_decode_avg_us',0)))
    except Exception as e:
        print(f, 'failed', e)
PY

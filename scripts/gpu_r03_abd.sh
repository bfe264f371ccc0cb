# same-box A/B of two library builds on the decomposed workloads (see gpu_r03_ab.sh)
cd $GRAFT_REPO_ROOT
Q="--workload decomp ${ARGS:---rank 250 --farnn 2} --steps ${STEPS:-100} --warmup 10 --no-cpu-baseline --no-pipelined --no-other-configs"
for rep in 1 2 3; do
for v in A B; do
  FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_var$v.so timeout 200 python bench.py $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v decomp', '%.2f us' % (d['ms_per_step']*1e3), d['parity']['tags_equal'])"
done
done

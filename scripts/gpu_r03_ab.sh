# Same-box A/B of two library builds (boxes differ by ~1 %: a change of that size is invisible across calls).  Build the candidate,
# copy it to re2nn-seq_amd/csrc/libfarnn_hip_varB.so, build the reference into ..._varA.so, then run this in ONE gpurun call.
cd $GRAFT_REPO_ROOT
Q="--steps 300 --warmup 30 --no-cpu-baseline --no-pipelined --no-other-configs"
for rep in 1 2 3; do
for v in A B; do
  FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_var$v.so timeout 200 python bench.py $Q 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v ragged', '%.2f us' % (d['ms_per_step']*1e3), d['parity']['tags_equal'])"
done
done
for v in A B; do
  FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_var$v.so timeout 200 python bench.py $Q --full-length 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v full-length', '%.2f us' % (d['ms_per_step']*1e3))"
  FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_var$v.so timeout 200 python bench.py $Q --workload ifst_crf 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v ifst_crf', '%.2f us' % (d['ms_per_step']*1e3))"
done

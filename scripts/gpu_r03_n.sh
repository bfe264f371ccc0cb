# Viterbi forward-step ablations (profiling build): which part of the step costs what
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03n; mkdir -p $O
Q="--steps 3 --warmup 2 --no-cpu-baseline --no-pipelined --no-other-configs --no-parity"
export FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_probes.so
for bits in 0 983040 2031616 4128768 3145728; do
  echo "== ablate $bits"
  FARNN_DBG=$((8192 + bits)) timeout 120 python bench.py --workload ifst_crf $Q 2>/dev/null | grep "^viterbi" | sort | tail -1 | sed 's/.*forward pass/forward pass/'
done | tee $O/ablate.txt

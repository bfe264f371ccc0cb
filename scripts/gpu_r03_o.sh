cd $GRAFT_REPO_ROOT
Q="--steps 3 --warmup 2 --no-cpu-baseline --no-pipelined --no-other-configs --no-parity"
FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_probes.so FARNN_DBG=8192 timeout 120 python bench.py --workload ifst_crf $Q > gpurun_out/o.log 2>&1; echo rc=$?
grep "^seq" gpurun_out/o.log | sort | tail -4 | cut -c1-200; grep "^viterbi" gpurun_out/o.log | sort | tail -3 | cut -c1-300

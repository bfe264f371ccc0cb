#!/bin/bash
# every SUPPORTED switch of include/farnn.h: the parity suites under it (results -> profiles/r05_switch_matrix.txt).
# Tests that assert WHICH kernel ran (name checks) are expected to fail under a switch that selects another one; the line says how
# many: everything else must pass.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05sw; mkdir -p $O; rm -f $O/*
T="tests/test_gpu_parity_onehot.py tests/test_gpu_parity_decomposed.py tests/test_gpu_parity_bench_size.py tests/test_gpu_chain_regs_shapes.py tests/test_gpu_chain_viterbi.py"
for sw in "" FARNN_NOFUSE=1 FARNN_FUSE=1 FARNN_NOREGS=1 FARNN_NODEST=1 FARNN_NOLABELMAP=1 FARNN_CV_ONE=1 FARNN_CV_STASH=1 FARNN_VITERBI_UNFUSED=1 FARNN_VITERBI_BP=1 FARNN_PREP=1 FARNN_NOSORT=1 FARNN_DECOMP_NOREGS=1 FARNN_ROWS_NOREGS=1 FARNN_ROWS_LPR4=1 FARNN_ROWS_LPR4=2 FARNN_WIDE_UNPAIRED=1; do
  lib=""; case "$sw" in FARNN_NODEST=1|FARNN_CV_ONE=1|FARNN_CV_STASH=1) lib="FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_probes.so FARNN_AB_CHILD=1";; esac   # (A/B-build forms)
  [ "$sw" = FARNN_CV_STASH=1 ] && sw="FARNN_CV_STASH=1 FARNN_CV_ONE=1"
  env $sw $lib timeout 900 python -m pytest $T -q -m gpu > $O/out.txt 2>&1
  r=$(grep -E "passed|failed" $O/out.txt | tail -1)
  f=$(grep -E "^FAILED" $O/out.txt | sed 's/ - .*//' | sed 's/^FAILED //' | tr '\n' ' ')
  echo "${sw:-default}: $r ${f:+[failed: $f]}" | tee -a $O/matrix.txt
done

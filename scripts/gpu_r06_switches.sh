#!/bin/bash
# every SUPPORTED switch of include/farnn.h: the parity suites under it -> gpurun_out/r06sw/matrix.txt (copied to
# profiles/r06_switch_matrix.txt), ONE LOG PER SETTING (gpurun_out/r06sw/<setting>.txt: round 5 reused one file and lost the
# log of its one red cell).  A failing setting's log tail is appended to matrix_failures.txt.
# Tests that assert WHICH kernel ran hold for the default dispatch only (tests/util.py: NO_SWITCH); everything else must pass.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06sw; mkdir -p $O; rm -f $O/*
T="tests/test_gpu_parity_onehot.py tests/test_gpu_parity_decomposed.py tests/test_gpu_parity_bench_size.py tests/test_gpu_chain_regs_shapes.py tests/test_gpu_chain_viterbi.py"
for sw in "" FARNN_NOFUSE=1 FARNN_FUSE=1 FARNN_NOREGS=1 FARNN_NODEST=1 FARNN_NOLABELMAP=1 FARNN_CV_ONE=1 FARNN_CV_STASH=1 FARNN_VITERBI_UNFUSED=1 FARNN_VITERBI_BP=1 FARNN_PREP=1 FARNN_NOSORT=1 FARNN_DECOMP_NOREGS=1 FARNN_ROWS_NOREGS=1 FARNN_ROWS_LPR4=1 FARNN_ROWS_LPR4=2 FARNN_ROWS_NOROUNDS=1 FARNN_WIDE_UNPAIRED=1 $EXTRA_SWITCHES; do
  lib=""; case "$sw" in FARNN_NODEST=1|FARNN_CV_ONE=1|FARNN_CV_STASH=1) lib="FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_probes.so FARNN_AB_CHILD=1";; esac   # (A/B-build forms)
  [ "$sw" = FARNN_CV_STASH=1 ] && sw="FARNN_CV_STASH=1 FARNN_CV_ONE=1"
  log=$O/$(echo "${sw:-default}" | tr ' =' '__').txt
  env $sw $lib timeout 900 python -m pytest $T -q -m gpu -p no:cacheprovider > $log 2>&1
  r=$(grep -E "passed|failed" $log | tail -1)
  f=$(grep -E "^FAILED" $log | sed 's/ - .*//' | sed 's/^FAILED //' | tr '\n' ' ')
  echo "${sw:-default}: $r ${f:+[failed: $f]}" | tee -a $O/matrix.txt
  [ -n "$f" ] && { echo "==== ${sw:-default}"; tail -80 $log; } >> $O/matrix_failures.txt
done
true

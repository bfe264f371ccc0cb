#!/bin/bash
# every SUPPORTED switch of include/farnn.h: the parity suites under it (results -> profiles/r04_switch_matrix.txt)
O=gpurun_out/r04sw; mkdir -p $O; rm -f $O/*
T="tests/test_gpu_parity_onehot.py tests/test_gpu_parity_decomposed.py tests/test_gpu_parity_bench_size.py tests/test_gpu_chain_regs_shapes.py tests/test_gpu_chain_viterbi.py"
for sw in "" FARNN_NOFUSE=1 FARNN_NOREGS=1 FARNN_NOLABELMAP=1 FARNN_CV_WIDE=1 FARNN_CV_STASH=1 FARNN_VITERBI_UNFUSED=1 FARNN_VITERBI_BP=1 FARNN_PREP=1 FARNN_NOSORT=1 FARNN_DECOMP_NOREGS=1 FARNN_DECOMP_OLD=1 FARNN_ROWS_NOREGS=1 FARNN_ROWS_LPR4=1 FARNN_ROWS_LPR4=2 FARNN_WIDE_UNPAIRED=1; do
  r=$(env $sw timeout 600 python -m pytest $T -q -m gpu 2>&1 | grep -E "passed|failed" | tail -1)
  echo "${sw:-default}: $r" | tee -a $O/matrix.txt
done

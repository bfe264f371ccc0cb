#!/bin/bash
# in-kernel phase probes (profiling build) of the kernels round 6 works on: the shipped configuration's rows kernel and its
# K = 75 score + Viterbi kernel (256 x 64 and the example's own 200 x 30), the rank-50 register kernel, config 4's Viterbi.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06_probes${1:+_$1}; mkdir -p $O; rm -f $O/*
export FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_probes.so
Z="--steps 3 --warmup 2 --no-cpu-baseline --no-pipelined --no-other-configs --no-parity"
S="--workload decomp --rank 250 --farnn 2 --crf"
FARNN_DBG=8192 timeout 120 python bench.py $S $Z 2>/dev/null | grep "^viterbi" | sort | tail -6 > $O/probe_viterbi_k75_b256_l64.txt
FARNN_DBG=8192 timeout 120 python bench.py $S --batch 200 --seqlen 30 --full-length $Z 2>/dev/null | grep "^viterbi" | sort | tail -6 > $O/probe_viterbi_k75_b200_l30_full_length.txt
rows_probe() {     # the set-up's parts (four launches) and step 8's phases, two samples per wavefront
  grep "^rows" > $O/rows_raw.txt
  grep "^rows launch" $O/rows_raw.txt | sort | uniq | head -2
  grep "round 0: set-up" $O/rows_raw.txt | sort | head -4
  grep "round 1: set-up" $O/rows_raw.txt | sort | head -4
  for w in 0 1 2 3 4 5 6 7; do grep "wave $w step 8" $O/rows_raw.txt | head -2; done
  rm -f $O/rows_raw.txt
}
FARNN_DBG=16 timeout 120 python bench.py $S $Z 2>&1 | rows_probe > $O/probe_decomp_rows_r250_b256_l64.txt
FARNN_DBG=16 timeout 120 python bench.py $S --batch 200 --seqlen 30 $Z 2>&1 | rows_probe > $O/probe_decomp_rows_r250_b200_l30.txt
FARNN_DBG=16 timeout 120 python bench.py --workload decomp --rank 150 --farnn 2 --crf --states 134 --batch 200 --seqlen 30 $Z 2>&1 | rows_probe > $O/probe_decomp_rows_r150_s134_b200_l30.txt
FARNN_DBG=4096 timeout 120 python bench.py --workload decomp $Z 2>/dev/null | grep "^regs" | sort | tail -8 > $O/probe_decomp_regs8_r50.txt
FARNN_DBG=8192 timeout 120 python bench.py --workload ifst_crf $Z 2>/dev/null | grep "^viterbi" | sort | tail -6 > $O/probe_viterbi_k130.txt
head -50 $O/*.txt

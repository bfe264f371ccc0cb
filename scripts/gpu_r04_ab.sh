#!/bin/bash
# same-box A/B: libfarnn_hip.so (A) against re2nn-seq_amd/csrc/libfarnn_hip_ab.so (B); usage: gpu_r04_ab.sh <bench args...>
O=gpurun_out/r04ab; mkdir -p $O; rm -f $O/*
B="python bench.py --no-cpu-baseline --no-other-configs --no-pipelined --steps 200 --warmup 20 $@"
for rep in 1 2 3; do
$B > $O/A_$rep.json 2>$O/A_$rep.err
FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_ab.so $B > $O/B_$rep.json 2>$O/B_$rep.err
done

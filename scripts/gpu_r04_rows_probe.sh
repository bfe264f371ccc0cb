#!/bin/bash
O=gpurun_out/r04p; mkdir -p $O
export FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_probes.so
B="python bench.py --no-cpu-baseline --no-other-configs --no-pipelined --no-parity --steps 2 --warmup 1 --workload decomp --farnn 2 --full-length --batch 128"
for r in 250 150; do
FARNN_DBG=16 $B --rank $r 2>&1 | grep "rows wg" | head -8 > $O/lpr8_r$r.txt
FARNN_ROWS_LPR4=1 FARNN_DBG=16 $B --rank $r 2>&1 | grep "rows wg" | head -8 > $O/lpr4_r$r.txt
done

#!/bin/bash
# ablations of decomp_rows_kernel at rank 250, farnn 2 (probes build: FARNN_DBG 1 = no products, 2 = no non-linearity, 4 = no stash stores, 8 = no prefetch)
O=gpurun_out/r04a; mkdir -p $O
export FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_probes.so
B="python bench.py --no-cpu-baseline --no-other-configs --no-pipelined --no-parity --steps 50 --warmup 5 --workload decomp --rank 250 --farnn 2"
for d in 0 1 2 4 8 13 15; do
  FARNN_DBG=$d $B > $O/lpr8_dbg$d.json 2>$O/lpr8_dbg$d.err
  FARNN_ROWS_LPR4=1 FARNN_DBG=$d $B > $O/lpr4_dbg$d.json 2>$O/lpr4_dbg$d.err
done
FARNN_DBG=0 $B --full-length > $O/lpr8_full.json 2>$O/lpr8_full.err
FARNN_ROWS_LPR4=1 FARNN_DBG=0 $B --full-length > $O/lpr4_full.json 2>$O/lpr4_full.err
FARNN_DBG=0 $B --full-length --batch 128 > $O/lpr8_full_b128.json 2>$O/lpr8_full_b128.err
FARNN_ROWS_LPR4=1 FARNN_DBG=0 $B --full-length --batch 128 > $O/lpr4_full_b128.json 2>$O/lpr4_full_b128.err

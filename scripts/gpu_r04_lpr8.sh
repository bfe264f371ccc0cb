#!/bin/bash
O=gpurun_out/r04l; mkdir -p $O; rm -f $O/*
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -15 > $O/pytest.txt
B="python bench.py --no-cpu-baseline --no-other-configs --no-pipelined --steps 100 --warmup 10"
for r in 250 150 120 100; do
  $B --workload decomp --rank $r --farnn 2 > $O/r${r}_f9.json 2>$O/r${r}_f9.err
  FARNN_ROWS_LPR4=1 $B --workload decomp --rank $r --farnn 2 > $O/r${r}_lpr4.json 2>$O/r${r}_lpr4.err
done
export FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_probes.so
B="python bench.py --no-cpu-baseline --no-other-configs --no-pipelined --no-parity --steps 2 --warmup 1 --workload decomp --farnn 2 --full-length --batch 128"
FARNN_DBG=16 $B --rank 250 2>&1 | grep "rows wg" | head -8 > $O/probe_f9_r250.txt

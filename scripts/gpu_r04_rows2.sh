#!/bin/bash
O=gpurun_out/r04l; mkdir -p $O; rm -f $O/*
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -6 > $O/pytest.txt
timeout 600 python tests/soak_decomp_shapes.py 200 2>&1 | tail -2 >> $O/pytest.txt
B="python bench.py --no-cpu-baseline --no-other-configs --no-pipelined --steps 100 --warmup 10"
for r in 250 150 100; do
  $B --workload decomp --rank $r --farnn 2 > $O/r${r}_f2.json 2>$O/r${r}_f2.err
  FARNN_ROWS_LPR4=1 $B --workload decomp --rank $r --farnn 2 > $O/r${r}_f2_lpr4.json 2>$O/r${r}_f2_lpr4.err
done
$B --workload decomp --rank 100 --farnn 1 > $O/r100_f1.json 2>$O/r100_f1.err
$B --workload decomp --rank 150 --farnn 2 --states 134 > $O/r150_f2_s134.json 2>$O/r150_f2_s134.err
export FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_probes.so
Z="--no-cpu-baseline --no-other-configs --no-pipelined --no-parity --steps 2 --warmup 1 --workload decomp --farnn 2 --full-length --batch 128"
FARNN_DBG=16 python bench.py $Z --rank 250 2>&1 | grep "rows wg" | head -8 > $O/probe_r250.txt

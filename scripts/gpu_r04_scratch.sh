#!/bin/bash
# round 4: the kernels whose scratch was retired -- parity first, then the timings that item named
set -x
O=gpurun_out/r04s; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -5 > $O/pytest.txt
B="python bench.py --no-cpu-baseline --no-other-configs --no-pipelined --steps 100 --warmup 10"
$B --workload decomp --rank 250 --farnn 2 > $O/decomp_r250_f2.json 2>$O/decomp_r250_f2.err
$B --workload decomp --rank 250 --farnn 2 --states 134 > $O/decomp_r250_f2_s134.json 2>$O/decomp_r250_f2_s134.err
$B --workload decomp --rank 150 --farnn 2 --states 134 > $O/decomp_r150_f2_s134.json 2>$O/decomp_r150_f2_s134.err
$B --workload train --rank 250 --farnn 2 --steps 20 --warmup 3 > $O/train_r250_f2.json 2>$O/train_r250_f2.err
$B --workload train --rank 250 --farnn 2 --batch 1024 --steps 10 --warmup 2 > $O/train_r250_f2_b1024.json 2>$O/train_r250_f2_b1024.err
FARNN_TRAIN_NSEQ=4 $B --workload train --rank 250 --farnn 2 --batch 1024 --steps 10 --warmup 2 > $O/train_r250_f2_b1024_ns4.json 2>$O/train_r250_f2_b1024_ns4.err
$B --workload train --rank 50 --farnn 0 --steps 20 --warmup 3 > $O/train_r50.json 2>$O/train_r50.err
$B --workload decomp > $O/decomp.json 2>$O/decomp.err
$B --workload ifst_crf > $O/ifst_crf.json 2>$O/ifst_crf.err
tail -2 $O/*.err | head -60

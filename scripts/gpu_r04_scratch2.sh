#!/bin/bash
set -x
O=gpurun_out/r04s2; mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -80 > $O/pytest.txt
B="python bench.py --no-cpu-baseline --no-other-configs --no-pipelined --steps 50 --warmup 5"
FARNN_ROWS_NOREGS=1 $B --workload decomp --rank 150 --farnn 2 --states 134 > $O/noregs_r150_s134.json 2>$O/noregs_r150_s134.err
$B --workload decomp --rank 150 --farnn 2 --states 134 > $O/r150_s134.json 2>$O/r150_s134.err

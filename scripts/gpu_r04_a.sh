#!/bin/bash
# round 4, first call: where the S = 104 onehot shapes stand on round 3's kernels (ring kernel + score launch)
mkdir -p gpurun_out/r04a
for s in 71 104 128; do
  python bench.py --workload ifst --states $s --steps 200 --warmup 20 --no-other-configs --no-cpu-baseline --no-pipelined > gpurun_out/r04a/ifst_$s.json 2> gpurun_out/r04a/ifst_$s.err
  python bench.py --workload ifst_crf --states $s --steps 200 --warmup 20 --no-cpu-baseline --no-pipelined > gpurun_out/r04a/crf_$s.json 2> gpurun_out/r04a/crf_$s.err
done
python bench.py --workload decomp --steps 400 --warmup 20 --no-cpu-baseline --no-pipelined > gpurun_out/r04a/decomp.json 2> gpurun_out/r04a/decomp.err
tail -c 600 gpurun_out/r04a/*.err

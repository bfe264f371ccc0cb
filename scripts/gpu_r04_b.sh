#!/bin/bash
# round 4: the wide form (72 < S <= 128) -- parity first, then its time
mkdir -p gpurun_out/r04b
timeout 900 python -m pytest tests/test_gpu_chain_regs_shapes.py tests/test_gpu_parity_onehot.py -x -q -m gpu > gpurun_out/r04b/pytest.log 2>&1
tail -15 gpurun_out/r04b/pytest.log
for s in 104 128 96; do
  timeout 300 python bench.py --workload ifst --states $s --steps 200 --warmup 20 --no-other-configs --no-cpu-baseline --no-pipelined > gpurun_out/r04b/ifst_$s.json 2> gpurun_out/r04b/ifst_$s.err
  tail -c 300 gpurun_out/r04b/ifst_$s.err
done

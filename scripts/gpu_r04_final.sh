#!/bin/bash
# final check at HEAD: the driver's round-end sequence (GPU tests, smoke, default bench in the driver's form)
O=gpurun_out/r04z; mkdir -p $O; rm -f $O/*
timeout 1800 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 > $O/pytest.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form.json 2>$O/bench.err
tail -1 $O/smoke.txt

#!/bin/bash
O=gpurun_out/r04d; mkdir -p $O
python bench.py --steps 20 --warmup 5 --cpu-seconds 2 > $O/default_20.json 2>$O/default_20.err
python bench.py --steps 20 --warmup 5 --cpu-seconds 2 > $O/default_20b.json 2>$O/default_20b.err

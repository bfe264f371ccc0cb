#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04k
rm -rf $O; mkdir -p $O
Q="--steps 200 --warmup 20 --no-other-configs --no-cpu-baseline --no-pipelined --no-parity"
for i in 1 2 3; do
python bench.py --workload ifst $Q > $O/ifst_dev_$i.json 2>/dev/null
FARNN_HOST_EPOCH=1 python bench.py --workload ifst $Q > $O/ifst_host_$i.json 2>/dev/null
FARNN_NOLABELMAP=1 python bench.py --workload ifst $Q > $O/ifst_nolm_dev_$i.json 2>/dev/null
FARNN_NOLABELMAP=1 FARNN_HOST_EPOCH=1 python bench.py --workload ifst $Q > $O/ifst_nolm_host_$i.json 2>/dev/null
done
python scripts/sumjson.py $O/*.json

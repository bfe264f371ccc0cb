#!/bin/bash
# round 5: the batch-dependent default (one launch while 2 B <= compute units, else recurrence + K2l): whole GPU suite, then the driver's form of both
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05d; rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-other-configs --no-pipelined"
timeout 150 $B --steps 20 --warmup 5 > $O/default_driver_0.json 2>$O/err0.txt || { echo "first launch failed or hung (rc $?)"; tail -5 $O/err0.txt; exit 1; }
timeout 3000 python -m pytest tests -m gpu -q --maxfail=10 > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
tail -8 $O/pytest.txt | cut -c1-200
for rep in 1 2 3; do
timeout 100 $B --steps 20 --warmup 5 > $O/default_driver_$rep.json 2>/dev/null
FARNN_FUSE=1 timeout 100 $B --steps 20 --warmup 5 > $O/fuse_driver_$rep.json 2>/dev/null
done
timeout 100 $B > $O/default_200.json 2>/dev/null
FARNN_FUSE=1 timeout 100 $B > $O/fuse_200.json 2>/dev/null
timeout 100 $B --batch 128 > $O/default_b128.json 2>/dev/null
FARNN_NOFUSE=1 timeout 100 $B --batch 128 > $O/two_b128.json 2>/dev/null
timeout 100 $B --batch 192 > $O/default_b192.json 2>/dev/null
FARNN_FUSE=1 timeout 100 $B --batch 192 > $O/fuse_b192.json 2>/dev/null
timeout 100 $B --graph 10 > $O/default_graph.json 2>/dev/null
python scripts/sumjson.py $O/*.json | cut -c1-200

#!/bin/bash
# same-box A/B: libfarnn_hip.so (A, the working tree) against re2nn-seq_amd/csrc/libfarnn_hip_prev.so (B, a build of another commit);
# a hang in A's first launch ends the script.  usage: gpu_ab.sh [pytest files...]
cd $GRAFT_REPO_ROOT
O=gpurun_out/ab; rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-other-configs --no-pipelined"
timeout 150 $B --steps 20 --warmup 5 > $O/A_driver_0.json 2>$O/err0.txt || { echo "first launch failed or hung (rc $?)"; tail -5 $O/err0.txt; exit 1; }
if [ $# -gt 0 ]; then timeout 1200 python -m pytest "$@" -x -q -m gpu 2>&1 | tail -3; fi
for rep in 1 2 3; do
timeout 100 $B --steps 20 --warmup 5 > $O/A_driver_$rep.json 2>/dev/null
FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_prev.so timeout 100 $B --steps 20 --warmup 5 > $O/B_driver_$rep.json 2>/dev/null
timeout 100 $B > $O/A_200_$rep.json 2>/dev/null
FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_prev.so timeout 100 $B > $O/B_200_$rep.json 2>/dev/null
done
FARNN_FUSE=1 timeout 100 $B > $O/A_fuse.json 2>/dev/null
FARNN_FUSE=1 FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_prev.so timeout 100 $B > $O/B_fuse.json 2>/dev/null
timeout 100 $B --batch 1024 > $O/A_b1024.json 2>/dev/null
FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_prev.so timeout 100 $B --batch 1024 > $O/B_b1024.json 2>/dev/null
python scripts/sumjson.py $O/*.json | cut -c1-200

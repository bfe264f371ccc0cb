#!/bin/bash
# round 5: every workgroup's life (FARNN_DBG=2048, profiling build, scripts/debug/wg_stamps.py) of the headline launch, destination
# split against FARNN_NODEST=1, one launch and two
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05b; rm -rf $O; mkdir -p $O
export FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_probes.so FARNN_DBG=2048
timeout 100 python scripts/debug/wg_stamps.py > $O/wg_dest.txt 2>&1
FARNN_NODEST=1 timeout 100 python scripts/debug/wg_stamps.py > $O/wg_nodest.txt 2>&1
FARNN_NOFUSE=1 timeout 100 python scripts/debug/wg_stamps.py > $O/wg_dest_nofuse.txt 2>&1
FARNN_NOFUSE=1 FARNN_NODEST=1 timeout 100 python scripts/debug/wg_stamps.py > $O/wg_nodest_nofuse.txt 2>&1
timeout 100 python scripts/debug/wg_stamps.py --full-length > $O/wg_dest_full.txt 2>&1
FARNN_NODEST=1 timeout 100 python scripts/debug/wg_stamps.py --full-length > $O/wg_nodest_full.txt 2>&1
tail -n 30 $O/*.txt

#!/bin/bash
cd $GRAFT_REPO_ROOT
P="--steps 3 --warmup 2 --no-cpu-baseline --no-pipelined --no-other-configs --no-parity"
FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_probes.so FARNN_DBG=1024 timeout 120 python bench.py $P 2>/dev/null | grep "^meet\|^finish\|^seq" | sort | awk 'NR%6==0' | tail -18

#!/bin/bash
# timing-only variants of the register-fed recurrence kernel: libfarnn_hip_ab<N>.so = the production objects with
# chain_regs.hip rebuilt under -DFARNN_ABLATE=<N> (bits: chain_regs.hip.h).  Select one with FARNN_LIB=<path>.
cd "$(dirname "$0")/../re2nn-seq_amd/csrc" || exit 1
python build.py > /dev/null || exit 1
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DFARNN_ABLATE=$n -c -o build/ab_chain_regs_$n.o chain_regs.hip || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libfarnn_hip_ab$n.so build/ab_chain_regs_$n.o $(ls build/*.o | grep -v "chain_regs\|ab_") || exit 1
done

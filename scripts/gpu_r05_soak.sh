#!/bin/bash
# round 5 soaks -> profiles/r05_soak.txt: random geometries of the decomposed kernels under the ONE float64 rule, of the register-fed
# recurrence (both launch forms, K2l, K1d) and of the CRF kernels, well beyond what the suite draws
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05soak; rm -rf $O; mkdir -p $O
PYTHONPATH=. timeout 2400 python tests/soak_decomp_shapes.py 1500 > $O/decomp_shapes.txt 2>&1; tail -4 $O/decomp_shapes.txt | cut -c1-300
FARNN_SHAPE_SOAK=800 timeout 1800 python -m pytest tests/test_gpu_chain_regs_shapes.py -q -m gpu 2>&1 | tail -2 | tee $O/chain_regs_shapes_800.txt
FARNN_SHAPE_SOAK=800 FARNN_FUSE=1 timeout 1800 python -m pytest tests/test_gpu_chain_regs_shapes.py -q -m gpu 2>&1 | tail -2 | tee $O/chain_regs_shapes_800_fuse.txt
FARNN_SHAPE_SOAK=500 timeout 1800 python -m pytest tests/test_gpu_chain_viterbi.py -q -m gpu 2>&1 | tail -2 | tee $O/chain_viterbi_500.txt
# (the one-launch CRF form lives in the A/B build: the same 500 draws there, all four forms of every draw)
FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_probes.so FARNN_AB_CHILD=1 FARNN_SHAPE_SOAK=500 timeout 1800 python -m pytest tests/test_gpu_chain_viterbi.py -q -m gpu 2>&1 | tail -2 | tee $O/chain_viterbi_500_ab_build.txt
PYTHONPATH=. timeout 1200 python tests/soak_crf_decomp.py 300 > $O/crf_decomp.txt 2>&1; tail -2 $O/crf_decomp.txt | cut -c1-300

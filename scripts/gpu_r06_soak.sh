#!/bin/bash
# round 6 soaks -> profiles/r06_soak.txt, on the round's final kernels: the rewritten Viterbi kernel (score phase on sixteen
# wavefronts, sliced back-trace, tail rule) over random tag counts / lengths / batch sizes far beyond what the suite draws -- every
# IB4 instantiation, label-map and dense output matrices, stash and LDS rows --, the decomposed kernels (rows set-up, branch-free
# non-linearity) under the ONE float64 rule, the register-fed recurrence under both launch forms.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06soak; rm -rf $O; mkdir -p $O
for seed in 11 22 33; do
  FARNN_SHAPE_SEED=$seed FARNN_SHAPE_SOAK=600 timeout 1800 python -m pytest tests/test_gpu_chain_viterbi.py -q -m gpu -p no:cacheprovider 2>&1 | tail -1 | sed "s/^/chain_viterbi shapes x 600, seed $seed (production library: two launches, FARNN_NOFUSE): /" | tee -a $O/summary.txt
done
FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_probes.so FARNN_AB_CHILD=1 FARNN_SHAPE_SOAK=500 timeout 1800 python -m pytest tests/test_gpu_chain_viterbi.py -q -m gpu -p no:cacheprovider 2>&1 | tail -1 | sed "s/^/chain_viterbi shapes x 500 (A\/B build: + the one-launch form): /" | tee -a $O/summary.txt
PYTHONPATH=. timeout 2400 python tests/soak_decomp_shapes.py 1500 > $O/decomp_shapes.txt 2>&1; tail -3 $O/decomp_shapes.txt | cut -c1-300 | tee -a $O/summary.txt
FARNN_SHAPE_SOAK=800 timeout 1800 python -m pytest tests/test_gpu_chain_regs_shapes.py -q -m gpu -p no:cacheprovider 2>&1 | tail -1 | sed "s/^/chain_regs shapes x 800 (default dispatch; every second label-map draw under FARNN_FUSE=1): /" | tee -a $O/summary.txt
PYTHONPATH=. timeout 1200 python tests/soak_crf_decomp.py 300 > $O/crf_decomp.txt 2>&1; tail -2 $O/crf_decomp.txt | cut -c1-300 | tee -a $O/summary.txt
timeout 1500 python -m pytest tests/test_gpu_handoff_soak.py -q -m gpu -p no:cacheprovider 2>&1 | tail -1 | sed "s/^/hand-off soak (tests\/test_gpu_handoff_soak.py, FARNN_FUSE=1 forms): /" | tee -a $O/summary.txt
for seed in 1 2 3; do PYTHONPATH=. timeout 1800 python -c "import sys; sys.path.insert(0, 'tests'); import soak_rows_rounds as s; sys.exit(s.run(1500, seed=$seed))" 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-300 | sed "s/^/seed $seed: /" | tee -a $O/summary.txt; done

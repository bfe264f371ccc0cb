#!/bin/bash
# round 5's switch matrix had ONE red cell: FARNN_NOREGS=1 -> tests/test_gpu_chain_viterbi.py::test_one_launch_form_in_the_ab_build.
# Re-run that cell N times (parent = production library, child = A/B build, as the matrix ran it), one log per run.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_noregs; mkdir -p $O; rm -f $O/*
for i in 1 2 3 4 5 6 7 8; do
  FARNN_NOREGS=1 timeout 600 python -m pytest tests/test_gpu_chain_viterbi.py -q -m gpu -p no:cacheprovider > $O/parent_$i.txt 2>&1
  echo "parent run $i: $(grep -E 'passed|failed' $O/parent_$i.txt | tail -1)" | tee -a $O/summary.txt
done
for i in 1 2 3 4; do
  FARNN_SHAPE_SEED=$((1000+i)) FARNN_SHAPE_SOAK=150 FARNN_NOREGS=1 FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_probes.so FARNN_AB_CHILD=1 timeout 800 python -m pytest tests/test_gpu_chain_viterbi.py -x -q -m gpu -p no:cacheprovider -k "not test_one_launch_form_in_the_ab_build" > $O/child_seed_$i.txt 2>&1
  echo "child (A/B build) seed $((1000+i)), 150 draws: $(grep -E 'passed|failed' $O/child_seed_$i.txt | tail -1)" | tee -a $O/summary.txt
done

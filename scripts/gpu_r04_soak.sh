#!/bin/bash
# round 4: the long soaks (evidence under profiles/r04_soak.txt): the one-launch step's hand-off at 20x the suite's launch counts
# (K1r and the wide form), random decomposed geometries, random CRF shapes, the training step's random shapes
O=gpurun_out/r04soak; mkdir -p $O; rm -f $O/*
{
echo "== hand-off soak, FARNN_SOAK_SCALE=20 (tests/test_gpu_handoff_soak.py: ~2.8e6 one-launch tagging steps, every launch's tags compared on the device)"
FARNN_SOAK_SCALE=20 timeout 1500 python -m pytest tests/test_gpu_handoff_soak.py -q -m gpu 2>&1 | grep -E "passed|failed|FAILED"
echo "== random decomposed geometries (tests/soak_decomp_shapes.py 1500)"
timeout 1200 python tests/soak_decomp_shapes.py 1500 2>&1 | tail -1
echo "== random CRF / decomposed shapes (tests/soak_crf_decomp.py)"
timeout 600 python tests/soak_crf_decomp.py 2>&1 | tail -2
echo "== one-launch hand-off, random shapes (tests/soak_fused_handoff.py)"
timeout 900 python tests/soak_fused_handoff.py 2>&1 | tail -2
} > $O/soak.txt 2>&1
cat $O/soak.txt

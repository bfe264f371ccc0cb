# Round-3 soaks at raised counts (the suite runs their short forms): hand-off under load, random geometries of the one-launch
# arg-max and CRF steps, random decomposed geometries.  gpurun --timeout 2400 -- 'bash scripts/gpu_soak_r03.sh'
cd $GRAFT_REPO_ROOT
O=gpurun_out/soak_r03; mkdir -p $O
FARNN_SOAK_SCALE=${SCALE:-3} timeout 600 python -m pytest tests/test_gpu_handoff_soak.py -m gpu -q --timeout=500 -p no:cacheprovider > $O/handoff.log 2>&1; echo "handoff rc=$? $(tail -1 $O/handoff.log)"
FARNN_SHAPE_SEED=${SEED:-31} FARNN_SHAPE_SOAK=${N:-20000} timeout 900 python -m pytest tests/test_gpu_chain_regs_shapes.py -m gpu -q --timeout=800 -p no:cacheprovider > $O/regs_shapes.log 2>&1; echo "regs shapes rc=$? $(tail -1 $O/regs_shapes.log)"
FARNN_SHAPE_SEED=${SEED:-31} FARNN_SHAPE_SOAK=${NV:-8000} timeout 900 python -m pytest tests/test_gpu_chain_viterbi.py -m gpu -q --timeout=800 -p no:cacheprovider > $O/cv_shapes.log 2>&1; echo "chain_viterbi shapes rc=$? $(tail -1 $O/cv_shapes.log)"
timeout 900 python tests/soak_decomp_shapes.py ${ND:-300} > $O/decomp_shapes.log 2>&1; echo "decomp shapes rc=$? $(tail -2 $O/decomp_shapes.log | tr '\n' ' ')"
grep -h "^E  \|FAILED\|Error" $O/*.log | head -10

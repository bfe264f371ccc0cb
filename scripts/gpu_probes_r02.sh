# In-kernel cycle probes quoted in DESIGN.md (K2, K2b, section 10): s_memtime phase counts printed by the kernels under FARNN_DBG.
cd $GRAFT_REPO_ROOT
O=gpurun_out/probes_r02
rm -rf $O; mkdir -p $O
Q="--steps 1 --warmup 1 --no-cpu-baseline --no-other-configs --no-parity"
python scripts/debug/vit_probe.py 2>&1 | grep -v amdgpu.ids > $O/viterbi_phases.txt
FARNN_DBG=32768 python bench.py --workload ifst $Q 2>&1 | grep "epilogue of" | head -12 > $O/fused_epilogue_timeline.txt
FARNN_DBG=16384 python bench.py --workload ifst $Q 2>&1 | grep "score tile" | head -8 > $O/score_tile_phases_fused_ifst.txt
FARNN_NOFUSE=1 FARNN_DBG=16384 python bench.py --workload ifst $Q 2>&1 | grep "score tile" | head -8 > $O/score_tile_phases_kernel_ifst.txt
FARNN_DBG=16384 python bench.py --workload decomp $Q 2>&1 | grep "score tile" | head -8 > $O/score_tile_phases_kernel_decomp.txt
FARNN_DBG=4096 python bench.py --workload decomp $Q 2>&1 | grep "regs kernel" | head -8 > $O/decomp_regs_phases.txt
for sh in "104 250 2" "104 150 2" "104 100 1" "134 150 2"; do python scripts/debug/rows_shapes.py $sh; FARNN_ROWS_NOREGS=1 python scripts/debug/rows_shapes.py $sh; done 2>&1 | grep -v amdgpu.ids > $O/rows_register_forms.txt
wc -l $O/*

# Round-3 in-kernel probes (profiling build: csrc/build.py --probes; the production library carries none of this):
# the headline kernel's workgroup timeline, its step phases and its tile phases -> gpurun_out/probes_r03/*.txt
cd $GRAFT_REPO_ROOT
O=gpurun_out/probes_r03
rm -rf $O; mkdir -p $O
export FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_probes.so
Q="--steps 3 --warmup 2 --no-cpu-baseline --no-pipelined --no-other-configs --no-parity"
timeout 120 python bench.py $Q 2>/dev/null | grep "^seq" | sort | tail -16 > $O/probe_chain_regs_timeline.txt
FARNN_DBG=256 timeout 120 python bench.py $Q 2>/dev/null | grep "chain phases" | sort | tail -8 > $O/probe_chain_regs_step_phases.txt
FARNN_DBG=512 timeout 120 python bench.py $Q 2>/dev/null | grep "all wavefronts" | sort | tail -8 > $O/probe_chain_regs_tile_phases.txt
FARNN_DBG=8192 FARNN_NOFUSE=1 timeout 120 python bench.py --workload ifst_crf $Q 2>/dev/null | grep "^viterbi" | sort | tail -4 > $O/probe_viterbi_phases.txt
FARNN_DBG=8192 timeout 120 python bench.py --workload ifst_crf $Q 2>/dev/null | grep "^viterbi\|^seq" | sort | tail -8 > $O/probe_chain_viterbi_phases.txt
FARNN_DBG=4096 FARNN_NOFUSE=1 timeout 120 python bench.py --workload decomp $Q 2>/dev/null | grep "^regs" | sort | tail -4 > $O/probe_decomp_regs8_phases.txt
wc -l $O/*.txt

# Round-5 profile: every number quoted in DESIGN.md / README.md comes from this script's outputs under profiles/r05_*.
#   gpurun --timeout 3000 -- 'bash scripts/gpu_profile_r05.sh'   then   python scripts/summarize_profile.py r05
# Bench lines, rocprofv3 kernel traces (--kernel-trace --stats) and PMC passes are SEPARATE runs (counters are never
# combined with trace domains; FETCH_SIZE and WRITE_SIZE in passes of their own, MI355X_MICROARCH.md).  The timed regions
# and the PMC passes rotate through four different batches (bench.py --batches 4, the default).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/prof_r05
rm -rf $O; mkdir -p $O
Q="--no-cpu-baseline --no-other-configs"
T="timeout 300"
AB="FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_probes.so"      # the A/B build: the forms the production library left behind (FARNN_NODEST, FARNN_CV_ONE)
$T python bench.py --steps 20 --warmup 5 2>/dev/null > $O/bench_default_driver_form.json
$T python bench.py 2>/dev/null > $O/bench_ifst.json
FARNN_FUSE=1 $T python bench.py --steps 20 --warmup 5 $Q 2>/dev/null > $O/bench_ifst_one_launch_driver_form.json
FARNN_FUSE=1 $T python bench.py $Q 2>/dev/null > $O/bench_ifst_one_launch.json
env $AB FARNN_FUSE=1 FARNN_NODEST=1 $T python bench.py $Q 2>/dev/null > $O/bench_ifst_one_launch_source_split_r04.json
env $AB FARNN_NODEST=1 $T python bench.py $Q 2>/dev/null > $O/bench_ifst_source_split_r04.json
FARNN_NOLABELMAP=1 FARNN_NOFUSE=1 $T python bench.py $Q 2>/dev/null > $O/bench_ifst_two_kernels_matrix_core_scores_r04.json
$T python bench.py --batches 1 $Q 2>/dev/null > $O/bench_ifst_one_batch_replayed.json
$T python bench.py --graph 10 $Q 2>/dev/null > $O/bench_ifst_graph_replay.json
$T python bench.py --full-length $Q 2>/dev/null > $O/bench_ifst_full.json
FARNN_FUSE=1 $T python bench.py --full-length $Q 2>/dev/null > $O/bench_ifst_full_one_launch.json
$T python bench.py --batch 1024 $Q 2>/dev/null > $O/bench_ifst_b1024.json
FARNN_FUSE=1 $T python bench.py --batch 1024 $Q 2>/dev/null > $O/bench_ifst_b1024_one_launch.json
$T python bench.py --batch 64 $Q 2>/dev/null > $O/bench_ifst_b64.json
FARNN_NOFUSE=1 $T python bench.py --batch 64 $Q 2>/dev/null > $O/bench_ifst_b64_two_kernels.json
$T python bench.py --workload ifst --states 104 $Q 2>/dev/null > $O/bench_ifst_s104.json
$T python bench.py --workload ifst --states 128 $Q 2>/dev/null > $O/bench_ifst_s128.json
$T python bench.py --workload ifst_crf $Q 2>/dev/null > $O/bench_ifst_crf.json
env $AB FARNN_NODEST=1 $T python bench.py --workload ifst_crf $Q 2>/dev/null > $O/bench_ifst_crf_source_split_r04.json
env $AB FARNN_CV_ONE=1 $T python bench.py --workload ifst_crf $Q 2>/dev/null > $O/bench_ifst_crf_one_launch.json
$T python bench.py --workload ifst_crf --states 104 $Q 2>/dev/null > $O/bench_ifst_crf_s104.json
env $AB FARNN_CV_ONE=1 $T python bench.py --workload ifst_crf --states 104 $Q 2>/dev/null > $O/bench_ifst_crf_s104_one_launch.json
$T python bench.py --workload decomp $Q --steps 300 2>/dev/null > $O/bench_decomp.json
$T python bench.py --workload decomp --rank 250 --farnn 2 $Q --steps 100 2>/dev/null > $O/bench_decomp_r250_farnn2.json
$T python bench.py --workload decomp --rank 250 --farnn 2 --crf $Q --steps 100 2>/dev/null > $O/bench_decomp_r250_farnn2_crf.json
$T python bench.py --workload decomp --rank 250 --farnn 2 --crf --batch 200 --seqlen 30 $Q --steps 100 2>/dev/null > $O/bench_decomp_r250_farnn2_crf_bz200_len30.json
$T python bench.py --workload decomp --rank 150 --farnn 2 --crf --states 134 --batch 200 --seqlen 30 $Q --steps 100 2>/dev/null > $O/bench_decomp_r150_farnn2_crf_s134_bz200_len30.json
$T python bench.py --workload decomp --rank 150 --farnn 2 $Q --steps 100 2>/dev/null > $O/bench_decomp_r150_farnn2.json
$T python bench.py --workload decomp1 $Q --steps 100 2>/dev/null > $O/bench_decomp1.json
$T python bench.py --workload decomp0 $Q --steps 100 2>/dev/null > $O/bench_decomp0.json
$T python bench.py --workload fst4 $Q --steps 20 --warmup 3 2>/dev/null > $O/bench_fst4.json
timeout 600 python bench.py --workload synth512 --batch 1024 --seqlen 128 --steps 3 --warmup 1 $Q 2>/dev/null > $O/bench_synth512.json
$T python bench.py --workload train --no-cpu-baseline --steps 50 2>/dev/null > $O/bench_train.json
$T python bench.py --workload train --rank 250 --farnn 2 --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null > $O/bench_train_r250_farnn2.json
$T python scripts/host_inclusive_rate.py 2>/dev/null | grep host-inclusive > $O/host_inclusive.txt
R="--no-cpu-baseline --no-other-configs --no-pipelined --no-parity --event-stride 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 200 --warmup 20 $R > $O/trace.log 2>&1
FARNN_FUSE=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_two -- python3 bench.py --steps 200 --warmup 20 $R > $O/trace_two.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_s104 -- python3 bench.py --workload ifst --states 104 --steps 200 --warmup 20 $R > $O/trace_s104.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_crf -- python3 bench.py --workload ifst_crf --steps 100 --warmup 10 $R > $O/trace_crf.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_decomp -- python3 bench.py --workload decomp --steps 100 --warmup 10 $R > $O/trace_decomp.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_decomp_r250 -- python3 bench.py --workload decomp --rank 250 --farnn 2 --crf --steps 100 --warmup 10 $R > $O/trace_decomp_r250.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_fst4 -- python3 bench.py --workload fst4 --steps 10 --warmup 2 $R > $O/trace_fst4.log 2>&1
P="--steps 20 --warmup 5 $R"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py $P > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py $P > $O/pmc_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_l2 -- python3 bench.py $P > $O/pmc_l2.log 2>&1
FARNN_FUSE=1 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_onelaunch -- python3 bench.py $P > $O/pmc_fetch_onelaunch.log 2>&1
FARNN_FUSE=1 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_onelaunch -- python3 bench.py $P > $O/pmc_write_onelaunch.log 2>&1
FARNN_FUSE=1 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_l2_onelaunch -- python3 bench.py $P > $O/pmc_l2_onelaunch.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_s104 -- python3 bench.py --workload ifst --states 104 $P > $O/pmc_fetch_s104.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_s104 -- python3 bench.py --workload ifst --states 104 $P > $O/pmc_write_s104.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_l2_s104 -- python3 bench.py --workload ifst --states 104 $P > $O/pmc_l2_s104.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_fst4 -- python3 bench.py --workload fst4 --steps 3 --warmup 1 $R > $O/pmc_fetch_fst4.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_synth512 -- python3 bench.py --workload synth512 --batch 1024 --seqlen 128 --steps 2 --warmup 1 $R > $O/pmc_fetch_synth512.log 2>&1
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $O/sq${i}_ifst -- python3 bench.py --workload ifst --steps 10 --warmup 3 $R > $O/sq${i}_ifst.log 2>&1
  FARNN_FUSE=1 rocprofv3 --pmc $grp --output-format csv -d $O/sq${i}_ifstonelaunch -- python3 bench.py --workload ifst --steps 10 --warmup 3 $R > $O/sq${i}_ifstonelaunch.log 2>&1
  FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_probes.so FARNN_FUSE=1 FARNN_NODEST=1 rocprofv3 --pmc $grp --output-format csv -d $O/sq${i}_ifstonelaunchsourcesplit -- python3 bench.py --workload ifst --steps 10 --warmup 3 $R > $O/sq${i}_ifstonelaunchsourcesplit.log 2>&1
done
# in-kernel probes (profiling build)
export FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_probes.so
Z="--steps 3 --warmup 2 --no-cpu-baseline --no-pipelined --no-other-configs --no-parity"
FARNN_DBG=32768 FARNN_FUSE=1 timeout 120 python bench.py $Z 2>/dev/null | grep "^seq" | sort | tail -16 > $O/probe_chain_regs_one_launch_timeline.txt
FARNN_FUSE=1 FARNN_DBG=1024 timeout 120 python bench.py $Z 2>/dev/null | grep "^finish\|^meet" | sort | tail -24 > $O/probe_finish_phases_one_launch.txt
FARNN_DBG=33024 timeout 120 python bench.py $Z 2>/dev/null | grep "chain phases" | sort | tail -8 > $O/probe_chain_regs_step_phases.txt
FARNN_FUSE=1 FARNN_DBG=33024 timeout 120 python bench.py $Z 2>/dev/null | grep "chain phases" | sort | tail -8 > $O/probe_chain_regs_step_phases_one_launch.txt
env $AB FARNN_DBG=33024 FARNN_NODEST=1 timeout 120 python bench.py $Z 2>/dev/null | grep "chain phases" | sort | tail -8 > $O/probe_chain_regs_step_phases_source_split_r04.txt
FARNN_DBG=32768 timeout 120 python bench.py $Z 2>/dev/null | grep "^seq" | sort | tail -8 > $O/probe_chain_regs_timeline.txt
FARNN_DBG=4096 timeout 120 python bench.py --full-length $Z 2>/dev/null | grep "^compact tag.*dir" | sed 's/seq [0-9]*/seq N/' | sort | uniq -c | sort -rn | head -16 > $O/probe_compact_tag.txt
FARNN_DBG=2048 timeout 100 python scripts/debug/ct_stamps.py 2>&1 | grep -v amdgpu.ids > $O/compact_tag_wg_lifetimes.txt
FARNN_DBG=2048 timeout 100 python scripts/debug/ct_stamps.py --full-length 2>&1 | grep -v amdgpu.ids > $O/compact_tag_wg_lifetimes_full_length.txt
env $AB FARNN_DBG=8192 FARNN_CV_ONE=1 timeout 120 python bench.py --workload ifst_crf $Z 2>/dev/null | grep "^viterbi\|^seq" | sort | tail -8 > $O/probe_chain_viterbi_phases.txt
FARNN_DBG=8192 timeout 120 python bench.py --workload ifst_crf $Z 2>/dev/null | grep "^viterbi\|^seq" | sort | tail -8 > $O/probe_recurrence_then_viterbi_phases.txt
FARNN_DBG=2048 FARNN_FUSE=1 timeout 100 python scripts/debug/wg_stamps.py 2>&1 | grep -v amdgpu.ids > $O/wg_lifetimes_one_launch.txt
env $AB FARNN_DBG=2048 FARNN_FUSE=1 FARNN_NODEST=1 timeout 100 python scripts/debug/wg_stamps.py 2>&1 | grep -v amdgpu.ids > $O/wg_lifetimes_one_launch_source_split_r04.txt
FARNN_DBG=2048 timeout 100 python scripts/debug/wg_stamps.py 2>&1 | grep -v amdgpu.ids > $O/wg_lifetimes_recurrence_only.txt
env $AB FARNN_DBG=2048 FARNN_NODEST=1 timeout 100 python scripts/debug/wg_stamps.py 2>&1 | grep -v amdgpu.ids > $O/wg_lifetimes_recurrence_only_source_split_r04.txt
FARNN_DBG=2048 FARNN_FUSE=1 timeout 100 python scripts/debug/wg_stamps.py --full-length 2>&1 | grep -v amdgpu.ids > $O/wg_lifetimes_one_launch_full_length.txt
unset FARNN_LIB
for pb in fastmath_ulp issue_rate ta_rate; do      # (built artefacts, not tracked: built here when the snapshot came without them)
  [ -x scripts/probe/$pb.bin ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I re2nn-seq_amd/csrc scripts/probe/$pb.hip -o scripts/probe/$pb.bin > /dev/null 2>&1
done
timeout 60 scripts/probe/fastmath_ulp.bin > $O/fastmath_ulp.txt 2>&1
timeout 100 scripts/probe/issue_rate.bin > $O/issue_rate.txt 2>&1
timeout 200 scripts/probe/ta_rate.bin > $O/ta_rate.txt 2>&1
timeout 300 python scripts/debug/pool_probe.py 2>&1 | grep -v amdgpu.ids > $O/pool_probe.txt
# keep only the small summaries (kernel_stats + counter collection), drop per-dispatch traces > 4 MB
find $O -name '*.csv' -size +4M -delete
find $O -name '*kernel_trace.csv' -delete
du -sh $O

# Round-6 profile: every number quoted in DESIGN.md / README.md comes from this script's outputs under profiles/r06_*.
#   git rev-parse HEAD > gpurun_out/prof_r06_head.txt   (the launcher stamps the committed tree it ships: scripts/run_profile_r06.sh)
#   gpurun --timeout 3000 -- 'bash scripts/gpu_profile_r06.sh'   then   python scripts/summarize_profile.py r06
# Bench lines, rocprofv3 kernel traces (--kernel-trace --stats) and PMC passes are SEPARATE runs (counters are never combined
# with trace domains; FETCH_SIZE and WRITE_SIZE in passes of their own, MI355X_MICROARCH.md).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/prof_r06
rm -rf $O; mkdir -p $O
Q="--no-cpu-baseline --no-other-configs"
T="timeout 300"
AB="FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_probes.so"      # the A/B build (FARNN_CV_ONE, FARNN_NODEST) and the in-kernel probes
last() { tail -1; }                                                 # (the result line is the last line of stdout)
$T python bench.py --steps 20 --warmup 5 2>/dev/null > $O/bench_default_driver_form_stdout.txt; cp gpurun_out/bench_full.json $O/bench_default_driver_form_full.json
$T python bench.py $Q 2>/dev/null | last > $O/bench_ifst.json
FARNN_FUSE=1 $T python bench.py $Q 2>/dev/null | last > $O/bench_ifst_one_launch.json
$T python bench.py --graph 10 $Q 2>/dev/null | last > $O/bench_ifst_graph_replay.json
$T python bench.py --full-length $Q 2>/dev/null | last > $O/bench_ifst_full.json
$T python bench.py --batch 1024 $Q 2>/dev/null | last > $O/bench_ifst_b1024.json
$T python bench.py --batch 64 $Q 2>/dev/null | last > $O/bench_ifst_b64.json
$T python bench.py --batch 200 --seqlen 30 $Q 2>/dev/null | last > $O/bench_ifst_b200_l30.json
$T python bench.py --workload ifst --states 104 $Q 2>/dev/null | last > $O/bench_ifst_s104.json
$T python bench.py --workload ifst_crf $Q 2>/dev/null | last > $O/bench_ifst_crf.json
env $AB FARNN_CV_ONE=1 $T python bench.py --workload ifst_crf $Q 2>/dev/null | last > $O/bench_ifst_crf_one_launch.json
$T python bench.py --workload ifst_crf --states 104 $Q 2>/dev/null | last > $O/bench_ifst_crf_s104.json
$T python bench.py --workload decomp $Q --steps 300 2>/dev/null | last > $O/bench_decomp.json
$T python bench.py --workload decomp --rank 250 --farnn 2 $Q --steps 100 2>/dev/null | last > $O/bench_decomp_r250_farnn2.json
$T python bench.py --workload decomp --rank 250 --farnn 2 --crf $Q --steps 100 2>/dev/null | last > $O/bench_decomp_r250_farnn2_crf.json
$T python bench.py --workload decomp --rank 250 --farnn 2 --crf --batch 200 --seqlen 30 $Q --steps 100 2>/dev/null | last > $O/bench_decomp_r250_farnn2_crf_bz200_len30.json
$T python bench.py --workload decomp --rank 150 --farnn 2 --crf --states 134 --batch 200 --seqlen 30 $Q --steps 100 2>/dev/null | last > $O/bench_decomp_r150_farnn2_crf_s134_bz200_len30.json
$T python bench.py --workload decomp --rank 100 --farnn 1 $Q --steps 100 2>/dev/null | last > $O/bench_decomp_r100_farnn1.json
$T python bench.py --workload decomp1 $Q --steps 100 2>/dev/null | last > $O/bench_decomp1.json
$T python bench.py --workload decomp0 $Q --steps 100 2>/dev/null | last > $O/bench_decomp0.json
$T python bench.py --workload fst4 $Q --steps 20 --warmup 3 2>/dev/null | last > $O/bench_fst4.json
timeout 600 python bench.py --workload synth512 --batch 1024 --seqlen 128 --steps 3 --warmup 1 $Q 2>/dev/null | last > $O/bench_synth512.json
$T python bench.py --workload train --no-cpu-baseline --steps 50 2>/dev/null | last > $O/bench_train.json
$T python bench.py --workload train --rank 250 --farnn 2 --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | last > $O/bench_train_r250_farnn2.json
$T python scripts/host_inclusive_rate.py 2>/dev/null | grep host-inclusive > $O/host_inclusive.txt
R="--no-cpu-baseline --no-other-configs --no-pipelined --no-parity --event-stride 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 200 --warmup 20 $R > $O/trace.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_s104 -- python3 bench.py --workload ifst --states 104 --steps 200 --warmup 20 $R > $O/trace_s104.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_crf -- python3 bench.py --workload ifst_crf --steps 100 --warmup 10 $R > $O/trace_crf.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_decomp -- python3 bench.py --workload decomp --steps 100 --warmup 10 $R > $O/trace_decomp.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_decomp_r250 -- python3 bench.py --workload decomp --rank 250 --farnn 2 --crf --steps 100 --warmup 10 $R > $O/trace_decomp_r250.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_decomp_r250_bz200 -- python3 bench.py --workload decomp --rank 250 --farnn 2 --crf --batch 200 --seqlen 30 --steps 100 --warmup 10 $R > $O/trace_decomp_r250_bz200.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_fst4 -- python3 bench.py --workload fst4 --steps 10 --warmup 2 $R > $O/trace_fst4.log 2>&1
P="--steps 20 --warmup 5 $R"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py $P > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py $P > $O/pmc_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_l2 -- python3 bench.py $P > $O/pmc_l2.log 2>&1
FARNN_FUSE=1 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_onelaunch -- python3 bench.py $P > $O/pmc_fetch_onelaunch.log 2>&1
FARNN_FUSE=1 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_onelaunch -- python3 bench.py $P > $O/pmc_write_onelaunch.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_s104 -- python3 bench.py --workload ifst --states 104 $P > $O/pmc_fetch_s104.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_s104 -- python3 bench.py --workload ifst --states 104 $P > $O/pmc_write_s104.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_l2_s104 -- python3 bench.py --workload ifst --states 104 $P > $O/pmc_l2_s104.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_fst4 -- python3 bench.py --workload fst4 --steps 3 --warmup 1 $R > $O/pmc_fetch_fst4.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_synth512 -- python3 bench.py --workload synth512 --batch 1024 --seqlen 128 --steps 2 --warmup 1 $R > $O/pmc_fetch_synth512.log 2>&1
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $O/sq${i}_decompr250crf -- python3 bench.py --workload decomp --rank 250 --farnn 2 --crf --steps 10 --warmup 3 $R > $O/sq${i}_decompr250crf.log 2>&1
  rocprofv3 --pmc $grp --output-format csv -d $O/sq${i}_ifstcrf -- python3 bench.py --workload ifst_crf --steps 10 --warmup 3 $R > $O/sq${i}_ifstcrf.log 2>&1
done
# in-kernel probes (profiling build)
bash scripts/gpu_r06_probes.sh final > $O/probes.log 2>&1
for f in gpurun_out/r06_probes_final/*.txt; do cp $f $O/$(basename $f); done
export FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_probes.so
Z="--steps 3 --warmup 2 --no-cpu-baseline --no-pipelined --no-other-configs --no-parity"
FARNN_DBG=33024 timeout 120 python bench.py $Z 2>/dev/null | grep "chain phases" | sort | tail -8 > $O/probe_chain_regs_step_phases.txt
FARNN_DBG=2048 timeout 100 python scripts/debug/wg_stamps.py 2>&1 | grep -v amdgpu.ids > $O/wg_lifetimes_recurrence_only.txt
unset FARNN_LIB
for pb in score_product pk_rate; do
  [ -x scripts/probe/$pb.bin ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I re2nn-seq_amd/csrc scripts/probe/$pb.hip -o scripts/probe/$pb.bin > /dev/null 2>&1
done
timeout 60 scripts/probe/score_product.bin 2>&1 | grep waves > $O/score_product.txt
timeout 60 scripts/probe/pk_rate.bin > $O/pk_rate.txt 2>&1
python scripts/gpu_r06_dispatch_grid.py 2>&1 | grep -v amdgpu.ids > $O/dispatch_grid.txt
find $O -name '*.csv' -size +4M -delete
find $O -name '*kernel_trace.csv' -delete
du -sh $O

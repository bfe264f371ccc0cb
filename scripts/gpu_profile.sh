# Round profile: bench line + rocprofv3 kernel trace + PMC passes (FETCH_SIZE / WRITE_SIZE / L2 hit) in
# separate runs, as MI355X_MICROARCH.md prescribes (counters never combined with trace domains).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/prof_r01
rm -rf $O; mkdir -p $O
python bench.py 2>/dev/null > $O/bench_ifst.json
python bench.py --full-length --no-cpu-baseline 2>/dev/null > $O/bench_ifst_full.json
python bench.py --workload ifst_crf --no-cpu-baseline 2>/dev/null > $O/bench_ifst_crf.json
python bench.py --workload decomp --no-cpu-baseline --steps 100 2>/dev/null > $O/bench_decomp.json
python bench.py --workload decomp --rank 100 --farnn 1 --no-cpu-baseline --steps 100 2>/dev/null > $O/bench_decomp_r100_farnn1.json
python bench.py --workload decomp --rank 250 --farnn 2 --no-cpu-baseline --steps 100 2>/dev/null > $O/bench_decomp_r250_farnn2.json
python bench.py --workload fst4 --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null > $O/bench_fst4.json
python bench.py --workload synth512 --batch 1024 --seqlen 128 --steps 3 --warmup 1 --no-cpu-baseline --event-stride 1 2>/dev/null > $O/bench_synth512.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-pipelined > $O/trace.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_crf -- python3 bench.py --workload ifst_crf --steps 100 --warmup 10 --no-cpu-baseline --no-pipelined > $O/trace_crf.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_decomp -- python3 bench.py --workload decomp --steps 100 --warmup 10 --no-cpu-baseline --no-pipelined > $O/trace_decomp.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pipelined > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pipelined > $O/pmc_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_l2 -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pipelined > $O/pmc_l2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_full -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pipelined --full-length > $O/pmc_fetch_full.log 2>&1
find $O -name '*.csv' | head -30
# keep only the small summaries (kernel_stats + counter collection), drop per-dispatch traces > 5 MB
find $O -name '*.csv' -size +5M -delete
du -sh $O
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_synth512 -- python3 bench.py --workload synth512 --batch 1024 --seqlen 128 --steps 2 --warmup 1 --no-cpu-baseline --event-stride 0 > $O/pmc_fetch_synth512.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_fst4 -- python3 bench.py --workload fst4 --steps 3 --warmup 1 --no-cpu-baseline --no-pipelined --event-stride 0 > $O/pmc_fetch_fst4.log 2>&1
find $O -name '*.csv' -size +5M -delete

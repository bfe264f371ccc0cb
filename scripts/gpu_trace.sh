cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/trace_tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/trace_tmp -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-pipelined "$@" > /dev/null 2>&1
head -6 gpurun_out/trace_tmp/*/*kernel_stats.csv | cut -c1-200
python bench.py --steps 300 --no-cpu-baseline --no-pipelined "$@" 2>/dev/null | cut -c1-250

# first bench + rocprofv3 kernel trace of the same command
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
python __graft_entry__.py smoke 2>&1 | tail -3
python bench.py --steps 200 --warmup 20 2>&1 | tee gpurun_out/bench_ifst.json
python bench.py --steps 200 --warmup 20 --full-length --no-cpu-baseline 2>&1 | tee gpurun_out/bench_ifst_full.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ifst -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/prof_ifst.log 2>&1
find gpurun_out/prof_ifst -name '*stats*' | head; 
f=$(find gpurun_out/prof_ifst -name '*kernel_stats.csv' | head -1); head -12 "$f"

cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for d in 0 1 2 3; do
FARNN_D1_DBG=$d rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/d1dbg$d -- python3 bench.py --workload decomp1 --steps 10 --warmup 3 --no-cpu-baseline --no-pipelined --event-stride 0 > gpurun_out/d1dbg$d.log 2>&1
f=$(ls gpurun_out/d1dbg$d/*/*kernel_stats.csv | head -1); echo "dbg $d: $(grep br_mfma $f | cut -d, -f2-4)"
find gpurun_out/d1dbg$d -name "*.csv" -size +1M -delete
done

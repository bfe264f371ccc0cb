cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "0 2" "1 2" "2 2" "3 2" "0 1"; do
set -- $cfg
FARNN_D1_DBG=$1 FARNN_D1_WGS=$2 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/d1dbg -- python3 bench.py --workload decomp1 --steps 10 --warmup 3 --no-cpu-baseline --no-pipelined --event-stride 0 > gpurun_out/d1dbg.log 2>&1
f=$(ls gpurun_out/d1dbg/*/*kernel_stats.csv | head -1); echo "dbg $1 wgs/cu $2: $(grep br_mfma $f | cut -d, -f2-4)"
rm -rf gpurun_out/d1dbg
done

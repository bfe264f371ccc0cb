cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/trace_crf
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --workload ifst_crf --steps 100 --warmup 10 --no-cpu-baseline --no-pipelined --event-stride 0 > $O/log.txt 2>&1
f=$(find $O -name '*_kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    print(r['Name'][:70], r['Calls'], r['AverageNs'], r['Percentage'])
PY
find $O -name '*.csv' -size +2M -delete

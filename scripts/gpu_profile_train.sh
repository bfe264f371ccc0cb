# rocprofv3 kernel stats for the training step (copied to profiles/) + full GPU suite.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/trainprof
rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --workload train --steps 50 --warmup 5 --no-cpu-baseline > $O/stats.log 2>&1
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
head -12 $O/kernel_stats.csv | cut -c1-150
rm -rf $O/stats

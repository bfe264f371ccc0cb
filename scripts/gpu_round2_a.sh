# round 2, first GPU pass: whole GPU suite + default bench line + N=2 self-spawn
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02a
rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
timeout 600 python bench.py > $O/bench_200.json 2> $O/bench_200.err; echo "bench200 rc=$?"
FARNN_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_n2.json 2> $O/bench_n2.err; echo "n2 rc=$?"
head -c 1500 $O/bench_default.json

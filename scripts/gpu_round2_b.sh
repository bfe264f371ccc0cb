# fused chain+score launch: parity (onehot suite, bench-size, soak) then timing with / without fusion and fences
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02b
rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity_onehot.py tests/test_gpu_parity_bench_size.py tests/test_gpu_fullsize_properties.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -15 $O/pytest.log
for v in "FARNN_NOFUSE=1" "FARNN_FUSE_FENCE=1" "FARNN_FUSE_FENCE=0"; do
  echo "== $v"
  env $v python bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['chain_avg_us'], d['roofline']['score_decode_avg_us'], d['roofline']['launches_timed'], d['pipelined']['ms_per_step'], d['parity']['tags_equal'])"
done
echo "== steps 20 (driver form)"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['launches_timed'], d['roofline']['frac'])"
FARNN_NOFUSE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('nofuse', d['value'], d['ms_per_step'], d['roofline']['launches_timed'], d['roofline']['frac'])"

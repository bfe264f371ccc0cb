#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04q
rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_chain_viterbi.py tests/test_gpu_parity_bench_size.py tests/test_gpu_parity_decomposed.py tests/test_gpu_parity_onehot.py -q -m gpu -x > $O/pytest.log 2>&1; tail -4 $O/pytest.log
Q="--steps 300 --warmup 20 --no-other-configs --no-cpu-baseline --no-pipelined"
for i in 1 2; do
python bench.py --workload ifst_crf $Q > $O/crf_71_$i.json 2>/dev/null
FARNN_NOFUSE=1 python bench.py --workload ifst_crf $Q > $O/crf_71_two_$i.json 2>/dev/null
done
python scripts/sumjson.py $O/*.json
P="--steps 3 --warmup 2 --no-cpu-baseline --no-pipelined --no-other-configs --no-parity"
FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_probes.so FARNN_DBG=8192 timeout 120 python bench.py --workload ifst_crf $P 2>/dev/null | grep "^viterbi\|^seq" | sort | tail -6

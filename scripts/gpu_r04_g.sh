#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04g
rm -rf $O; mkdir -p $O
python scripts/debug/r04_flat.py 2>&1 | grep -v amdgpu.ids | grep -v "n 0 flat" | tail -20
echo "--- with FARNN_NOLABELMAP=1"
FARNN_NOLABELMAP=1 python scripts/debug/r04_flat.py 2>&1 | grep -v amdgpu.ids | grep -v "n 0 flat" | tail -20
timeout 1500 python -m pytest tests/test_gpu_chain_regs_shapes.py tests/test_gpu_parity_onehot.py tests/test_gpu_parity_bench_size.py tests/test_gpu_chain_viterbi.py tests/test_gpu_parity_decomposed.py -q -m gpu > $O/pytest.log 2>&1; tail -8 $O/pytest.log
Q="--steps 200 --warmup 20 --no-other-configs --no-cpu-baseline --no-pipelined"
python bench.py --workload ifst $Q > $O/ifst_71.json 2>/dev/null
python bench.py --workload ifst --states 104 $Q > $O/ifst_104.json 2>/dev/null
python bench.py --workload ifst_crf $Q > $O/crf_71.json 2>/dev/null
FARNN_NOFUSE=1 python bench.py --workload ifst_crf $Q > $O/crf_71_two.json 2>/dev/null
python bench.py --workload ifst_crf --states 104 $Q > $O/crf_104.json 2>/dev/null
python bench.py --workload decomp $Q > $O/decomp.json 2>/dev/null
python scripts/sumjson.py $O/*.json
P="--steps 3 --warmup 2 --no-cpu-baseline --no-pipelined --no-other-configs --no-parity"
FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_probes.so timeout 120 python bench.py $P 2>/dev/null | grep "^seq" | sort | tail -6
FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_probes.so FARNN_DBG=8192 timeout 120 python bench.py --workload ifst_crf $P 2>/dev/null | grep "^viterbi\|^seq" | sort | tail -6

#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04h
rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity_bench_size.py tests/test_gpu_chain_viterbi.py -q -m gpu -x > $O/pytest.log 2>&1; tail -5 $O/pytest.log
Q="--steps 200 --warmup 20 --no-other-configs --no-cpu-baseline --no-pipelined"
for i in 1 2; do
python bench.py --workload ifst $Q > $O/ifst_71_$i.json 2>/dev/null
FARNN_NOLABELMAP=1 python bench.py --workload ifst $Q > $O/ifst_71_nolm_$i.json 2>/dev/null
python bench.py --workload decomp $Q > $O/decomp_$i.json 2>/dev/null
FARNN_NOLABELMAP=1 python bench.py --workload decomp $Q > $O/decomp_nolm_$i.json 2>/dev/null
python bench.py --workload ifst_crf $Q > $O/crf_71_$i.json 2>/dev/null
FARNN_NOLABELMAP=1 python bench.py --workload ifst_crf $Q > $O/crf_71_nolm_$i.json 2>/dev/null
done
python scripts/sumjson.py $O/*.json
P="--steps 3 --warmup 2 --no-cpu-baseline --no-pipelined --no-other-configs --no-parity"
FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_probes.so FARNN_DBG=1024 timeout 120 python bench.py $P 2>/dev/null | grep "^finish" | sort | tail -6
FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_probes.so FARNN_DBG=1024 FARNN_NOLABELMAP=1 timeout 120 python bench.py $P 2>/dev/null | grep "^finish" | sort | tail -4
FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_probes.so FARNN_DBG=8192 timeout 120 python bench.py --workload ifst_crf $P 2>/dev/null | grep "^viterbi\|^seq" | sort | tail -6

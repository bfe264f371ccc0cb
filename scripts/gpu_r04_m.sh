#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04m
rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; tail -5 $O/pytest.log
Q="--steps 200 --warmup 20 --no-other-configs --no-cpu-baseline --no-pipelined"
python bench.py --workload ifst_crf --states 104 $Q > $O/crf_104.json 2>/dev/null
FARNN_NOFUSE=1 python bench.py --workload ifst_crf --states 104 $Q > $O/crf_104_two.json 2>/dev/null
python bench.py --workload ifst_crf $Q > $O/crf_71.json 2>/dev/null
FARNN_NOFUSE=1 python bench.py --workload ifst_crf $Q > $O/crf_71_two.json 2>/dev/null
python bench.py --workload ifst $Q > $O/ifst_71.json 2>/dev/null
python scripts/sumjson.py $O/*.json
P="--steps 3 --warmup 2 --no-cpu-baseline --no-pipelined --no-other-configs --no-parity"
FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_probes.so FARNN_DBG=8192 timeout 120 python bench.py --workload ifst_crf --states 104 $P 2>/dev/null | grep "^viterbi\|^seq" | sort | tail -6

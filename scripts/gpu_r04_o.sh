#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04o
rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; tail -5 $O/pytest.log
Q="--steps 200 --warmup 20 --no-other-configs --no-cpu-baseline --no-pipelined"
for i in 1 2; do
python bench.py --workload ifst $Q > $O/ifst_71_$i.json 2>/dev/null
FARNN_NOLABELMAP=1 python bench.py --workload ifst $Q > $O/ifst_71_nolm_$i.json 2>/dev/null
python bench.py --workload ifst --states 104 $Q > $O/ifst_104_paired_$i.json 2>/dev/null
FARNN_WIDE_UNPAIRED=1 python bench.py --workload ifst --states 104 $Q > $O/ifst_104_unpaired_$i.json 2>/dev/null
FARNN_NOFUSE=1 python bench.py --workload ifst --states 104 $Q > $O/ifst_104_two_$i.json 2>/dev/null
done
python bench.py --workload ifst --states 96 $Q > $O/ifst_96.json 2>/dev/null
python bench.py --workload ifst --states 128 $Q > $O/ifst_128.json 2>/dev/null
python bench.py --workload ifst --full-length $Q > $O/ifst_71_full.json 2>/dev/null
python bench.py --workload ifst --steps 20 --warmup 5 --no-other-configs --no-cpu-baseline --no-pipelined > $O/ifst_71_driver.json 2>/dev/null
python scripts/sumjson.py $O/*.json
P="--steps 3 --warmup 2 --no-cpu-baseline --no-pipelined --no-other-configs --no-parity"
FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_probes.so FARNN_DBG=1024 timeout 120 python bench.py $P 2>/dev/null | grep "^finish\|^meet" | sort | tail -6
FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_probes.so FARNN_DBG=1024 timeout 120 python bench.py $P 2>/dev/null | grep "^seq" | sort | tail -4
FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_probes.so FARNN_DBG=1024 timeout 120 python bench.py --workload ifst --states 104 $P 2>/dev/null | grep "^seq" | sort | tail -4

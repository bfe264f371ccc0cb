#!/bin/bash
# round 4: the wide form at S = 104 -- full GPU suite, the workgroups' timeline (profiling build), fabric traffic (PMC passes)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04c
rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; tail -5 $O/pytest.log
Q="--steps 3 --warmup 2 --no-cpu-baseline --no-pipelined --no-other-configs --no-parity"
FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_probes.so timeout 120 python bench.py --workload ifst --states 104 $Q 2>/dev/null | grep "^seq" | sort | tail -16 > $O/probe_wide104_timeline.txt
FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_probes.so FARNN_DBG=512 timeout 120 python bench.py --workload ifst --states 104 $Q 2>/dev/null | grep "all wavefronts" | sort | tail -8 > $O/probe_wide104_tile_phases.txt
cat $O/probe_wide104_timeline.txt $O/probe_wide104_tile_phases.txt
R="--no-cpu-baseline --no-other-configs --no-pipelined --no-parity --event-stride 0"
P="--workload ifst --states 104 --steps 20 --warmup 5 $R"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 bench.py $P > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 bench.py $P > $O/pmc_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_l2 -- python3 bench.py $P > $O/pmc_l2.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 bench.py --workload ifst --states 104 --steps 200 --warmup 20 $R > $O/trace.log 2>&1
FARNN_NOFUSE=1 python bench.py --workload ifst --states 104 --steps 200 --warmup 20 --no-other-configs --no-cpu-baseline --no-pipelined > $O/ifst_104_nofuse.json 2>/dev/null
python bench.py --workload ifst --states 104 --steps 200 --warmup 20 --no-other-configs --no-cpu-baseline > $O/ifst_104.json 2>/dev/null
python bench.py --workload ifst_crf --states 104 --steps 200 --warmup 20 --no-other-configs --no-cpu-baseline --no-pipelined > $O/crf_104.json 2>/dev/null
find $O -name '*.csv' -size +4M -delete
find $O -name '*kernel_trace.csv' -delete
python - <<'PY'
import csv, glob, collections
for sub in ('pmc_fetch','pmc_write','pmc_l2'):
    d=collections.defaultdict(lambda:[0.0,0])
    for f in glob.glob('gpurun_out/r04c/%s/*/*_counter_collection.csv'%sub):
        for row in csv.DictReader(open(f)):
            k=(row['Kernel_Name'].split('(')[0][:60],row['Counter_Name'])
            d[k][0]+=float(row['Counter_Value']); d[k][1]+=1
    for k,(s,n) in sorted(d.items()):
        if 'chain' in k[0]: print(sub,k,s/n,n)
PY

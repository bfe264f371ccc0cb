#!/bin/bash
# round 4: full GPU suite with the reference-captured bench-size fixtures; scaling experiments of the wide form
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04d
rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest.log 2>&1; tail -8 $O/pytest.log
Q="--workload ifst --states 104 --steps 200 --warmup 20 --no-other-configs --no-cpu-baseline --no-pipelined --no-parity"
python bench.py $Q --batch 128 > $O/w104_b128.json 2>/dev/null
python bench.py $Q --full-length > $O/w104_full.json 2>/dev/null
FARNN_NOFUSE=1 python bench.py $Q --batch 128 > $O/w104_b128_nofuse.json 2>/dev/null
FARNN_NOFUSE=1 python bench.py $Q --full-length > $O/w104_full_nofuse.json 2>/dev/null
python scripts/sumjson.py $O/*.json

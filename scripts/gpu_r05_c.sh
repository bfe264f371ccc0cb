#!/bin/bash
# round 5: the one-launch compact tagging kernel (compact_tag.hip.h): guard, parity, the bench's `compact` line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05c; rm -rf $O; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity_onehot.py -x -q -m gpu -k "compact or atis_scale" > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
tail -5 $O/pytest.txt
B="python bench.py --no-cpu-baseline --no-other-configs --no-pipelined --steps 200 --warmup 20"
timeout 150 $B > $O/A_0.json 2>$O/A_0.err
FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_probes.so FARNN_DBG=4096 timeout 100 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pipelined --no-other-configs --no-parity 2>/dev/null | grep "^compact tag" | sort | uniq -c | sort -rn | head -12
FARNN_NOFUSE=1 timeout 150 $B > $O/A_nofuse.json 2>$O/A_nofuse.err
python - <<'PY'
import json
for f in ('A_0', 'A_nofuse'):
    try:
        d = json.loads(open('gpurun_out/r05c/%s.json' % f).read().strip().splitlines()[-1])
        print(f, 'dense %.2f us' % (d['ms_per_step'] * 1e3), 'compact:', {k: (round(v, 3) if isinstance(v, float) else v) for k, v in d.get('compact', {}).items() if k != 'note'})
    except Exception as e:
        print(f, 'ERR', e)
PY

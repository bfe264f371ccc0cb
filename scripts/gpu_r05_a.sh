#!/bin/bash
# round 5, the destination-split compute wavefronts (chain_dest.hip.h): a guarded first launch (a hang ends the script), parity,
# same-box A/B against FARNN_NODEST=1, probes.  Worst case of every step summed stays under the gpurun limit.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05a; rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-other-configs --no-pipelined --steps 200 --warmup 20"
timeout 150 $B > $O/A_0.json 2>$O/A_0.err || { echo "first launch failed or hung (rc $?)"; tail -5 $O/A_0.err; exit 1; }
timeout 600 python -m pytest tests/test_gpu_parity_onehot.py tests/test_gpu_chain_regs_shapes.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
for rep in 1 2 3; do
timeout 100 $B > $O/A_$rep.json 2>$O/A_$rep.err
FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_probes.so FARNN_NODEST=1 timeout 100 $B > $O/B_$rep.json 2>$O/B_$rep.err
done
FARNN_NOFUSE=1 timeout 100 $B > $O/A_nofuse.json 2>$O/A_nofuse.err
FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_probes.so FARNN_NOFUSE=1 FARNN_NODEST=1 timeout 100 $B > $O/B_nofuse.json 2>$O/B_nofuse.err
timeout 100 $B --full-length > $O/A_full.json 2>$O/A_full.err
FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_probes.so FARNN_NODEST=1 timeout 100 $B --full-length > $O/B_full.json 2>$O/B_full.err
export FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_probes.so
Q="--steps 3 --warmup 2 --no-cpu-baseline --no-pipelined --no-other-configs --no-parity"
timeout 100 python bench.py $Q 2>/dev/null | grep "^seq" | sort | tail -16 > $O/probe_chain_regs_timeline.txt
FARNN_DBG=256 timeout 100 python bench.py $Q 2>/dev/null | grep "chain phases" | sort | tail -8 > $O/probe_chain_regs_step_phases.txt
FARNN_NOFUSE=1 timeout 100 python bench.py $Q 2>/dev/null | grep "^seq" | sort | tail -8 > $O/probe_chain_regs_nofuse_timeline.txt
python scripts/sumjson.py $O/*.json > $O/summary.txt 2>&1
tail -3 $O/pytest.txt; cat $O/summary.txt; cat $O/probe_chain_regs_step_phases.txt; tail -4 $O/probe_chain_regs_timeline.txt; tail -4 $O/probe_chain_regs_nofuse_timeline.txt

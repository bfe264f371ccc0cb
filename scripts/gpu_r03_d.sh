cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03d
O=gpurun_out/r03d
FARNN_DBG=512 FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_probes.so timeout 120 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-pipelined --no-other-configs --no-parity > $O/p.json 2> $O/p.err
grep "^seq" $O/p.json | cut -c1-200 | tail -12

# cache-policy variants of K1d's ring loads (libfarnn_hip_v_*.so built with -DFARNN_RD_CPOL) against the shipped library, same box
cd $GRAFT_REPO_ROOT
O=gpurun_out/cpol; rm -rf $O; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-other-configs --no-pipelined"
timeout 150 $B --steps 20 --warmup 5 > $O/base_0.json 2>$O/err0.txt || { echo "first launch failed or hung (rc $?)"; tail -5 $O/err0.txt; exit 1; }
for rep in 1 2; do
  timeout 100 $B > $O/base_200_$rep.json 2>/dev/null
  for v in nt sc0 sc1 sc0sc1; do
    FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_v_$v.so timeout 100 $B > $O/${v}_200_$rep.json 2>/dev/null
  done
done
timeout 100 $B --full-length > $O/base_full.json 2>/dev/null
for v in nt sc0 sc1 sc0sc1; do
  FARNN_LIB=$PWD/re2nn-seq_amd/csrc/libfarnn_hip_v_$v.so timeout 100 $B --full-length > $O/${v}_full.json 2>/dev/null
done
python scripts/sumjson.py $O/*.json | cut -c1-200

cd $GRAFT_REPO_ROOT
for B in 256 255 128 64; do echo "== B=$B"; python scripts/debug/decomp_bench_size.py 50 0 $B 2>&1 | tail -8; done
echo "== NOSORT"; FARNN_NOSORT=1 python scripts/debug/decomp_bench_size.py 50 0 256 2>&1 | tail -4
echo "== PREP"; FARNN_PREP=1 python scripts/debug/decomp_bench_size.py 50 0 256 2>&1 | tail -4
echo "== NSEQ2"; FARNN_ROWS_NSEQ=2 python scripts/debug/decomp_bench_size.py 50 0 256 2>&1 | tail -4
echo "== OLD"; FARNN_DECOMP_OLD=1 python scripts/debug/decomp_bench_size.py 50 0 256 2>&1 | tail -4

#!/bin/bash
# round 6's inner loop: the parity suites that cover the kernels being worked on, the in-kernel probes, and three bench lines
cd $GRAFT_REPO_ROOT
T="tests/test_gpu_parity_bench_size.py tests/test_gpu_chain_viterbi.py tests/test_gpu_parity_decomposed.py tests/test_gpu_parity_onehot.py"
timeout 1200 python -m pytest $T -q -m gpu -p no:cacheprovider -x 2>&1 | tail -8
bash scripts/gpu_r06_probes.sh $1 > gpurun_out/r06_probes_$1.log 2>&1
for f in gpurun_out/r06_probes_$1/probe_viterbi*.txt gpurun_out/r06_probes_$1/probe_decomp_rows_r250*.txt gpurun_out/r06_probes_$1/probe_decomp_regs8_r50.txt; do echo "== $f"; sort $f | uniq | head -4; done
for w in "--workload decomp --rank 250 --farnn 2 --crf" "--workload decomp --rank 250 --farnn 2 --crf --batch 200 --seqlen 30" "--workload decomp --rank 150 --farnn 2 --crf --states 134 --batch 200 --seqlen 30" "--workload ifst_crf" "--workload decomp" "--workload decomp --rank 250 --farnn 2"; do
  python bench.py $w --steps 300 --warmup 20 --no-cpu-baseline --no-other-configs --no-pipelined 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', '| ms/step %.4f chain %.1f score %.1f' % (d['ms_per_step'], d['roofline']['chain_avg_us'], d['roofline']['score_decode_avg_us']), d['parity'])"
done

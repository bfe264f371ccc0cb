cd $GRAFT_REPO_ROOT
pr() { python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1 tok/s %.3g ms/step %.4f chain_us %.1f frac %.3f achieved %.0f GB/s score_us %.1f' % (d['value'], d['ms_per_step'], r['kernel_avg_us'], r['frac'], r['achieved'], r['score_decode_avg_us']))"; }
python bench.py --workload synth512 --batch 256 --seqlen 64 --steps 5 --warmup 2 --no-cpu-baseline --event-stride 1 2>gpurun_out/synth512.err | pr "synth512 B256 L64"
tail -3 gpurun_out/synth512.err
python bench.py --workload synth512 --batch 1024 --seqlen 128 --steps 3 --warmup 1 --no-cpu-baseline --event-stride 1 2>gpurun_out/synth512b.err | pr "synth512 B1024 L128"
tail -3 gpurun_out/synth512b.err

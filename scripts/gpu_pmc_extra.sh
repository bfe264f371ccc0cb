# Extra PMC passes (one counter group per run, never combined with trace domains): LDS bank conflicts and
# instruction mix of the four kernels the round's claims are about.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/pmc_extra
rm -rf $O; mkdir -p $O
i=0
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  for wl in ifst ifst_crf decomp; do
    rocprofv3 --pmc $grp --output-format csv -d $O/g${i}_$wl -- python3 bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --no-pipelined --event-stride 0 > $O/g${i}_$wl.log 2>&1
  done
done
python3 - <<'PY'
import csv, glob, collections, os
rows = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob('gpurun_out/pmc_extra/*/*/*_counter_collection.csv'):
    wl = f.split('/')[2].split('_', 1)[1]
    for r in csv.DictReader(open(f)):
        k = (wl, r['Kernel_Name'].split('(')[0].replace('void ', ''), r['Counter_Name'])
        rows[k][0] += float(r['Counter_Value']); rows[k][1] += 1
with open('gpurun_out/pmc_extra/summary.csv', 'w') as out:
    out.write('workload,kernel,counter,mean_per_dispatch,dispatches\n')
    for (wl, kn, c), (s, n) in sorted(rows.items()):
        if n >= 3 and any(t in kn for t in ('chain_kernel', 'score_tile', 'viterbi', 'decomp_rows')):
            out.write('{},"{}",{},{:.1f},{}\n'.format(wl, kn, c, s / n, n))
print(open('gpurun_out/pmc_extra/summary.csv').read())
PY
find $O -name '*.csv' -size +1M -delete

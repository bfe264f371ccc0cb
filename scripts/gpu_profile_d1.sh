# rocprofv3 kernel stats + PMC passes for the decomposed independent=1 workload (results copied to profiles/).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/d1prof
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --workload decomp1 --steps 50 --warmup 5 --no-cpu-baseline --no-pipelined --event-stride 0 > $O/stats.log 2>&1
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $O/g${i} -- python3 bench.py --workload decomp1 --steps 6 --warmup 2 --no-cpu-baseline --no-pipelined --event-stride 0 > $O/g${i}.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
rows = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob('gpurun_out/d1prof/g*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = (r['Kernel_Name'].split('(')[0].replace('void ', ''), r['Counter_Name'])
        rows[k][0] += float(r['Counter_Value']); rows[k][1] += 1
with open('gpurun_out/d1prof/pmc_summary.csv', 'w') as out:
    out.write('kernel,counter,mean_per_dispatch,dispatches\n')
    for (kn, c), (s, n) in sorted(rows.items()):
        if n >= 3 and ('decomp1' in kn or 'chain_kernel' in kn):
            out.write('"{}",{},{:.1f},{}\n'.format(kn, c, s / n, n))
print(open('gpurun_out/d1prof/pmc_summary.csv').read())
PY
head -6 $O/kernel_stats.csv | cut -c1-160
rm -rf $O/stats $O/g1 $O/g2 $O/g3 $O/g4 $O/g5

# PMC passes for the decomposed independent=1 scoring kernel (MFMA utilisation and wait reasons).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/pmc_d1
rm -rf $O; mkdir -p $O
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $O/g${i} -- python3 bench.py --workload decomp1 --steps 6 --warmup 2 --no-cpu-baseline --no-pipelined --event-stride 0 > $O/g${i}.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
rows = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob('gpurun_out/pmc_d1/*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = (r['Kernel_Name'].split('(')[0].replace('void ', ''), r['Counter_Name'])
        rows[k][0] += float(r['Counter_Value']); rows[k][1] += 1
for (kn, c), (s, n) in sorted(rows.items()):
    if 'decomp1' in kn or 'chain' in kn:
        print('{:60s} {:36s} {:16.1f} {}'.format(kn[:60], c, s / n, n))
PY
tail -3 $O/g1.log
find $O -name '*.csv' -size +1M -delete

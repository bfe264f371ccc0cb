#!/usr/bin/env python3
"""Fold gpurun_out/prof_rNN (bench lines, rocprofv3 kernel stats, PMC passes) into the small,
committed summaries under profiles/.  Usage: python scripts/summarize_profile.py r01"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, 'gpurun_out', 'prof_' + tag)
dst = os.path.join(ROOT, 'profiles')
os.makedirs(dst, exist_ok=True)

for f in glob.glob(os.path.join(src, 'bench_*.json')) + glob.glob(os.path.join(src, 'bench_*_stdout.txt')):
    if os.path.getsize(f) > 0:
        shutil.copy(f, os.path.join(dst, '{}_{}'.format(tag, os.path.basename(f))))
for f in [f_ for f_ in glob.glob(os.path.join(src, '*.txt')) if not f_.endswith('_stdout.txt')]:
    # (a probe's printf lines and the bench's JSON line share stdout: keep the probe text only)
    keep = []
    for ln in open(f, errors='replace').read().splitlines():
        if '{"metric"' in ln:
            ln = ln.split('{"metric"')[0].rstrip()
            if len(ln) < 40:
                continue
            ln += ' [...]'
        keep.append(ln)
    with open(os.path.join(dst, '{}_{}'.format(tag, os.path.basename(f))), 'w') as g:
        g.write('\n'.join(keep) + '\n')
for sub, name in (('trace', 'ifst'), ('trace_two', 'ifst_one_launch'), ('trace_crf', 'ifst_crf'), ('trace_decomp', 'decomp'),
                  ('trace_fst4', 'fst4'), ('trace_s104', 'ifst_s104'), ('trace_crf_s104', 'ifst_crf_s104'), ('trace_decomp_r250', 'decomp_r250_farnn2'),
                  ('trace_decomp_r250_bz200', 'decomp_r250_farnn2_crf_bz200_len30')):
    ks = sorted(glob.glob(os.path.join(src, sub, '*', '*_kernel_stats.csv')), key=os.path.getmtime)
    if ks:
        shutil.copy(ks[-1], os.path.join(dst, '{}_{}_kernel_stats.csv'.format(tag, name)))


def agg(path):
    d = collections.defaultdict(lambda: [0.0, 0])
    files = sorted(glob.glob(os.path.join(src, path, '*', '*_counter_collection.csv')), key=os.path.getmtime)
    if not files:
        return {}
    for row in csv.DictReader(open(files[-1])):       # gpurun merges runs: take the newest

        k = (row['Kernel_Name'].split('(')[0].replace('void ', ''), row['Counter_Name'])
        d[k][0] += float(row['Counter_Value'])
        d[k][1] += 1
    return {k: (s / n, n) for k, (s, n) in d.items()}


rows = []
runs = [('pmc_fetch', 'ifst ragged U[5,64]'), ('pmc_write', 'ifst ragged U[5,64]'), ('pmc_l2', 'ifst ragged U[5,64]'),
        ('pmc_fetch_onelaunch', 'ifst ragged U[5,64] one launch (FARNN_FUSE=1)'), ('pmc_write_onelaunch', 'ifst ragged U[5,64] one launch (FARNN_FUSE=1)'),
        ('pmc_l2_onelaunch', 'ifst ragged U[5,64] one launch (FARNN_FUSE=1)'),
        ('pmc_fetch_full', 'ifst full-length'), ('pmc_fetch_synth512', 'synth512 B1024 L128'),
        ('pmc_fetch_fst4', 'fst4'), ('pmc_fetch_s104', 'ifst S=104 ragged U[5,64]'), ('pmc_write_s104', 'ifst S=104 ragged U[5,64]'),
        ('pmc_l2_s104', 'ifst S=104 ragged U[5,64]')]
for run, label in runs:
    for (k, c), (mean, n) in sorted(agg(run).items()):
        if 'farnn::' in k and any(t in k for t in ('chain', 'score', 'viterbi', 'decomp_regs', 'decomp_rows')):
            rows.append([label, k, c, '%.1f' % mean, n])
with open(os.path.join(dst, tag + '_pmc_summary.csv'), 'w') as f:
    w = csv.writer(f)
    w.writerow(['workload', 'kernel', 'counter', 'mean_per_dispatch', 'dispatches'])
    w.writerows(rows)


def pick(label, kernel, counter):
    for r in rows:
        if r[0] == label and kernel in r[1] and r[2] == counter:
            return float(r[3])
    return None


traffic = {}
try:                                     # entries this profile did not re-measure keep their own stamp
    with open(os.path.join(dst, 'traffic.json')) as f:
        traffic = {k: v for k, v in json.load(f).items() if not k.startswith('_')}
except (OSError, ValueError):
    pass
measured_now = []
note = ('(2*FETCH_SIZE + WRITE_SIZE) * 1024 per dispatch from separate rocprofv3 --pmc passes; on gfx950 '
        'FETCH_SIZE tallies 128-B requests at 64 B for 16-B/lane streams, hence the factor 2 '
        '(MI355X_MICROARCH.md, HBM section)')
f_, w_ = pick('ifst ragged U[5,64]', 'chain', 'FETCH_SIZE'), pick('ifst ragged U[5,64]', 'chain', 'WRITE_SIZE')
if f_ is not None:
    measured_now.append('ifst')
    traffic['ifst'] = {'hbm_bytes_per_launch': (2 * f_ + (w_ or 0)) * 1024, 'kernel': 'chain_regs_kernel', 'source': note}
f_, w_ = pick('ifst ragged U[5,64] one launch (FARNN_FUSE=1)', 'chain', 'FETCH_SIZE'), pick('ifst ragged U[5,64] one launch (FARNN_FUSE=1)', 'chain', 'WRITE_SIZE')
if f_ is not None:
    measured_now.append('ifst_one_launch')
    traffic['ifst_one_launch'] = {'hbm_bytes_per_launch': (2 * f_ + (w_ or 0)) * 1024, 'kernel': 'chain_regs_kernel<fused>', 'source': note}
f_, w_ = pick('ifst S=104 ragged U[5,64]', 'chain', 'FETCH_SIZE'), pick('ifst S=104 ragged U[5,64]', 'chain', 'WRITE_SIZE')
if f_ is not None:
    measured_now.append('ifst_s104')
    traffic['ifst_s104'] = {'hbm_bytes_per_launch': (2 * f_ + (w_ or 0)) * 1024, 'kernel': 'chain_wide_kernel', 'source': note}
f_ = pick('synth512 B1024 L128', 'chain', 'FETCH_SIZE')
if f_ is not None:
    measured_now.append('synth512')
    traffic['synth512'] = {'hbm_bytes_per_launch': 2 * f_ * 1024, 'kernel': 'chain_kernel', 'source': note + ' (reads only)'}
f_ = pick('fst4', 'fst4_score', 'FETCH_SIZE')
if f_ is not None:
    measured_now.append('fst4')
    traffic['fst4'] = {'hbm_bytes_per_launch': 2 * f_ * 1024, 'kernel': 'fst4_score_kernel', 'source': note + ' (reads only)'}
# SQ counter groups (instruction mix, waits, LDS conflicts) of the dominant kernels
sq = collections.defaultdict(lambda: [0.0, 0])
for d in sorted(glob.glob(os.path.join(src, 'sq*_*'))):
    if not os.path.isdir(d):
        continue
    wl = os.path.basename(d).split('_', 1)[1]
    files = sorted(glob.glob(os.path.join(d, '*', '*_counter_collection.csv')), key=os.path.getmtime)
    if not files:
        continue
    for row in csv.DictReader(open(files[-1])):
        kn = row['Kernel_Name'].split('(')[0].replace('void ', '')
        if 'farnn::' in kn and any(t in kn for t in ('chain_kernel', 'chain_regs', 'chain_wide', 'score_tile', 'viterbi', 'decomp_regs', 'decomp_rows')):
            key = (wl, kn, row['Counter_Name'])
            sq[key][0] += float(row['Counter_Value']); sq[key][1] += 1
with open(os.path.join(dst, tag + '_pmc_sq.csv'), 'w') as f:
    w = csv.writer(f)
    w.writerow(['workload', 'kernel', 'counter', 'mean_per_dispatch', 'dispatches'])
    for (wl, kn, c), (s_, n) in sorted(sq.items()):
        w.writerow([wl, kn, c, '%.1f' % (s_ / n), n])
# the split is valid for the kernels of the commit it was measured at: stamp it (bench.py prints it as traffic_head)
import subprocess
stamp = os.path.join(ROOT, 'gpurun_out', 'prof_{}_head.txt'.format(tag))
if os.path.exists(stamp):        # written by the launcher (scripts/run_profile_r06.sh) from the tree it shipped: "<head>[+dirty]"
    head, dirty = open(stamp).read().strip(), False
else:
    try:
        head = subprocess.run(['git', 'rev-parse', '--short=12', 'HEAD'], cwd=ROOT, capture_output=True, text=True).stdout.strip()
        dirty = bool(subprocess.run(['git', 'status', '--porcelain', '--', 're2nn-seq_amd/csrc'], cwd=ROOT, capture_output=True, text=True).stdout.strip())
    except Exception:
        head, dirty = 'unknown', False
for k_ in traffic:
    if traffic[k_].get('source', '').startswith('(2*FETCH') and k_ in measured_now:
        traffic[k_]['head'] = head + ('+dirty' if dirty else '')
traffic['_measured_at'] = {'head': head + ('+dirty' if dirty else ''), 'profile': tag, 'entries': measured_now}
with open(os.path.join(dst, 'traffic.json'), 'w') as f:
    json.dump(traffic, f, indent=1)
for r in rows:
    print(r)
print(json.dumps(traffic, indent=1))

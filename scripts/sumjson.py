#!/usr/bin/env python3
"""print one line per bench JSON file: python scripts/sumjson.py gpurun_out/x/*.json"""
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, 'ERR', e); continue
    def line(tag, d):
        r = d.get('roofline', {})
        par = d.get('parity', {})
        print('%-34s %8.2f us/step %.3e tok/s | %-28s chain %.1f score %.1f frac %.3f %s | parity %s' % (
            tag, d['ms_per_step'] * 1e3, d['value'], r.get('kernel', '')[:28], r.get('chain_avg_us', 0), r.get('score_decode_avg_us', 0),
            r.get('frac', 0) or 0, r.get('bound', ''), par.get('tags_equal')))
    line(f.split('/')[-1], d)
    for o in d.get('other_configs', []) or []:
        line('  + ' + str(o.get('workload')), o)
    if 'pipelined' in d and d['pipelined']:
        print('      pipelined:', {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d['pipelined'].items() if k in ('value', 'ms_per_step', 'streams')})

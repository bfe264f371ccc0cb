"""The bench line's size contract (CPU-only): round 5's ONE result line grew to 20 KB and the round driver could not parse
it (BENCH_r05.json: "parsed": null).  The line is now built by bench.result_line() from the full result dict and must
stay <= 4 KB whatever the measurement carries; each side measurement is its own short `{"other_config": ...}` line.
The canned input is the full 20 KB result of round 5's default invocation (profiles/r05_bench_default_driver_form.json)."""
import copy
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402  (imports torch; touches no GPU at import)


def _canned():
    with open(os.path.join(ROOT, 'profiles', 'r05_bench_default_driver_form.json')) as f:
        return json.load(f)


def _check_line(line, n_gpus):
    assert '\n' not in line and len(line) <= 4096, len(line)
    d = json.loads(line)
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline'):
        assert k in d, k
    assert d['n_gpus'] == n_gpus and d['vs_baseline'] is None
    rf = d['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'kernel_avg_us', 'kernel'):
        assert k in rf, k
    assert 0 < rf['frac'] and abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-3 * rf['frac']
    assert 'workload' in d['config'] and 'model' not in d['config']
    return d


def test_default_line_is_small_and_complete():
    full = _canned()
    assert len(json.dumps(full)) > 15000                      # the input really is round 5's oversized result
    line = bench.result_line(full)
    d = _check_line(line, 1)
    assert len(line) <= 3000, len(line)                       # today's line: well under the cap
    assert d['cpu_baseline']['value'] > 0 and d['cpu_baseline']['cores'] >= 1 and d['cpu_baseline']['kind'] == 'port'
    assert len(d['cpu_baseline']['sample']) <= 160
    assert d['parity']['tags_equal'] is True
    # one-number summaries of the side measurements; the other configs by label -> ms per step
    assert d['compact']['value'] > 0 and d['pipelined']['value'] > 0 and d['host_inclusive']['value'] > 0
    assert set(d['other_configs_ms_per_step']) == {o['workload'] for o in full['other_configs']}
    assert all(isinstance(v, float) and v > 0 for v in d['other_configs_ms_per_step'].values())
    assert d['other_configs_parity'] is True
    assert 'other_configs' not in d and 'peak_note' not in d['roofline'] and 'note' not in d['roofline']


def test_other_config_lines_are_short():
    full = _canned()
    total = 0
    for o in full['other_configs']:
        line = bench.other_config_line(o)
        assert len(line) <= bench.OTHER_LINE_CAP and not line.startswith('{"metric"'), (len(line), line)
        d = json.loads(line)
        assert d['other_config'] == o['workload'] and d['value'] > 0 and d['ms_per_step'] > 0
        assert d['parity']['tags_equal'] is True and d['roofline']['frac'] > 0
        total += len(line) + 1
    # all of them and the result line fit the 8 KB tail the driver keeps of stdout
    assert total + len(bench.result_line(full)) < 8000, total
    err = bench.other_config_line({'workload': 'x', 'error': 'RuntimeError: ' + 'y' * 5000})
    assert len(err) <= bench.OTHER_LINE_CAP and json.loads(err)['other_config'] == 'x'


def test_multi_rank_line_same_schema_and_cap():
    full = _canned()
    for k in ('other_configs', 'compact', 'cpu_baseline', 'cpu_baseline_faithful', 'host_inclusive', 'pipelined'):
        full.pop(k)
    full['n_gpus'] = 8
    full['gather'] = {'backend': 'farnn_rccl_gather_tags (ncclAllGather, libfarnn_rccl.so, side HIP stream)', 'rccl_version_code': 22203,
                      'comm_count': 8}
    d = _check_line(bench.result_line(full), 8)
    assert d['gather']['comm_count'] == 8 and 'cpu_baseline' not in d


def test_cap_holds_when_the_result_grows():
    """whatever a later round adds to the full result, the line sheds its optional summaries before it breaks the cap"""
    full = _canned()
    big = copy.deepcopy(full)
    big['other_configs'] = [dict(o, workload='{}_{}'.format(o['workload'], i)) for i in range(12) for o in full['other_configs']]
    big['roofline']['peak_note'] = 'x' * 10000
    big['config']['workload'] = 'w' * 5000
    line = bench.result_line(big)
    d = _check_line(line, 1)
    assert 'other_configs_ms_per_step' not in d and d['cpu_baseline']['value'] > 0


def test_no_fraction_above_one_when_the_split_model_is_beaten():
    """roofline rule: a cache-resident kernel that beats its modelled split is priced at the L2 gather rate
    for every byte; the split reading stays beside it."""
    full = _canned()
    rf = full['roofline']
    # round 5's line was written under the old rule (frac 1.03 with model_falsified); rebuild it under the new one
    if rf.get('model_falsified') and rf['frac'] > 1:
        rf.update(peak_split=rf['peak'], frac_split=rf['frac'], peak=bench.L2_GATHER_GBS, frac=rf['achieved'] / bench.L2_GATHER_GBS)
    d = json.loads(bench.result_line(full))
    assert d['roofline']['frac'] <= 1.0 and d['roofline']['model_falsified'] is True
    assert d['roofline']['frac_split'] > 1.0 and abs(d['roofline']['frac'] - d['roofline']['frac_all_l2']) < 1e-6

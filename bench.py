#!/usr/bin/env python3
"""Benchmark of the FA-RNN forward tagging path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload ifst|ifst_crf|fst4|decomp|...]

A "step" is one pass of the hot path (farnn_tag through the C-ABI: recurrence + score/decode,
plus the RCCL gather of tag ids when N > 1) over one batch of synthetic input already resident
in HBM.  Default workload = BASELINE.json configs[1]: ATIS-BIO-sized onehot i-FST (V=950, S=71,
C=128), batch 256 x seqlen 64 per GPU, lengths ~ U[5,64] with one full-length row (BASELINE.md
section 3).  Weak scaling: every rank tags its own 256-sequence shard; `value` = valid (non-pad)
tokens tagged by all ranks per second.

`--gpus N` with N > 1 and no WORLD_SIZE in the environment: this process (which never touches the
GPU) starts the N ranks itself (`python -m torch.distributed.run --nproc-per-node N bench.py ...`),
relays rank 0's JSON line and exits with the children's status.  Under an external torchrun
(WORLD_SIZE set) the process is one of the ranks.

Rank 0 prints ONE JSON line (contract in the task statement), carrying
  roofline      the dominant kernel of the workload against the ceiling that really bounds it (see
                roofline_for): HBM stream, Infinity-Cache/L2 row gather, f32 VALU or f32 MFMA; its mean
                duration is measured with HIP events on the launch stream inside the timed region;
  cpu_baseline  the C port of the oracle (oracle/farnn_oracle.c, OpenMP over sequence-directions, T+W
                hoisted: the "fair" CPU number) on this host's cores, same batch (N=1, rank 0 only);
  cpu_baseline_faithful   the numpy restatement of the reference's own algorithm (re-adds T+W every call,
                walks all L pad steps, batched einsum like model_onehot.py:366-403);
  parity        the GPU tags (and scores, for the float paths) of the batch that was timed, re-checked
                against the oracle AFTER the timed region, for whichever workload ran;
  other_configs (default invocation only) the other single-GPU BASELINE configs -- ifst_crf, decomp
                (rank 50), fst4 -- and the decomposed model in the shape of the reference's shipped example
                configurations (rank 250, farnn 2), timed for a few hundred ms each with their own roofline and parity.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import numpy as np  # noqa: E402
import torch  # noqa: E402

# ceilings (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0            # HBM3E spec peak; ~6300 achievable with a float4 copy
IC_GATHER_GBS = 8600.0           # "Indexed rows: gather into LDS": 38 MB table, random rows, Infinity Cache
L2_GATHER_GBS = 16800.0          # same section: rows served by the XCD's L2 (lower figure of 16.8-18.8 TB/s)
INFINITY_CACHE_BYTES = 256 << 20
F32_PEAK_TFLOPS = 157.3          # dense f32 peak, vector and matrix alike (64 FLOP/clk/SIMD)

WORKLOADS = {
    # name: (description, V, S, C)
    'ifst': ('ATIS-BIO-sized onehot i-FST (--method onehot --independent 2)', 950, 71, 128),
    'ifst_crf': ('ATIS-BIO-sized onehot i-FST + fused Viterbi decode (use_crf=1)', 950, 71, 128),
    'fst4': ('ATIS-BIO-sized onehot FST, dense T[V,C,S,S] (--independent 0)', 950, 71, 128),
    'decomp': ('SNIPS-BIO-sized decomposed i-FST (--method decompose --independent 2)', 11000, 104, 73),
    'decomp1': ('SNIPS-BIO-sized decomposed independent=1 (FARNN_S_D_W_I), output rank 70', 11000, 104, 73),
    'decomp0': ('SNIPS-BIO-sized decomposed independent=0 (FARNN_S_D_W), wildcard rank 70', 11000, 104, 73),
    # one training step (forward with stash, cross-entropy, BPTT, gradient reductions, Adam)
    'train': ('SNIPS-BIO-sized decomposed i-FST, TRAINING step (farnn 0, tanh, CE1 loss, Adam)', 11000, 104, 73),
    # BASELINE configs[4], one GPU's shard: i-FST layout only (the 4-D layout would be 5.5 PB)
    'synth512': ('synthetic onehot i-FST V=20k S=512 C=256 (T = 21 GB fp32 per GPU, + transposed copy)',
                 20000, 512, 256),
}
BASE_STATES = {k: v[2] for k, v in WORKLOADS.items()}      # (before --states edits the table)
# the other single-GPU BASELINE configs the default invocation also times: (workload, steps, warmup)
# (workload, steps, warmup, argument overrides, label): the last decomposed entry is the shape of the reference's shipped example
# configurations (model_seq/example/*.res: --rank 250 --farnn 2)
# (an 'env' override: library environment switches for that run only -- the CRF step's two-launch form beside its one-launch default)
OTHER_CONFIGS = (('ifst_crf', 2000, 50, {}, 'ifst_crf'),
                 # the onehot path at the state count of the reference's SNIPS-BIO / ATIS-ZH-BIO automata (RE.py:56-60): the wide form
                 ('ifst', 1000, 50, {'states': 104}, 'ifst_s104'), ('ifst_crf', 600, 30, {'states': 104}, 'ifst_crf_s104'),
                 ('decomp', 1500, 50, {}, 'decomp'),
                 ('decomp', 400, 20, {'rank': 250, 'farnn': 2}, 'decomp_r250_farnn2'),
                 # the configuration the reference ships results for (all six model_seq/example/*.res: --rank 250 / 150 --farnn 2
                 # --use_crf 1 --bz 200 --seq_max_len 30; two of them --additional_states 30): at the example's own batch shape
                 # and at the benchmark's
                 ('decomp', 300, 20, {'rank': 250, 'farnn': 2, 'crf': True}, 'decomp_r250_farnn2_crf'),
                 ('decomp', 300, 20, {'rank': 250, 'farnn': 2, 'crf': True, 'batch': 200, 'seqlen': 30}, 'decomp_r250_farnn2_crf_bz200_len30'),
                 ('decomp', 200, 10, {'rank': 150, 'farnn': 2, 'crf': True, 'states': 134, 'batch': 200, 'seqlen': 30},
                  'decomp_r150_farnn2_crf_s134_bz200_len30'),
                 ('fst4', 60, 5, {}, 'fst4'),
                 # one GPU's shard of BASELINE configs[4] (V = 20 k, S = 512, C = 256; 1 024 sequences x 128 positions: 42 GB of blocks,
                 # generated on the device): the HBM-streamed end of the path, in the driver's own line since round 5
                 ('synth512', 3, 1, {'batch': 1024, 'seqlen': 128}, 'synth512_shard_b1024_len128'))


NATIVE_COMM = [None]        # --gather native: this rank's communicator (re2nn_seq_amd._rccl.Communicator)


# ------------------------------------------------------------------------------------------ the result line
LINE_CAP = 4096            # bytes: the ONE result line (round 5's grew to 20 KB and the driver could not parse it)
OTHER_LINE_CAP = 640       # bytes: each `{"other_config": ...}` line printed before it
FULL_PATH = os.path.join('gpurun_out', 'bench_full.json')


def _sig(v, n=6):
    """floats to n significant digits, recursively (the line is read by people and parsers, not diffed bit for bit)"""
    if isinstance(v, bool) or v is None:
        return v
    if isinstance(v, float):
        return float('{:.{}g}'.format(v, n)) if v == v and abs(v) != float('inf') else None
    if isinstance(v, dict):
        return {k: _sig(x, n) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_sig(x, n) for x in v]
    return v


def _pick(d, keys):
    return {k: d[k] for k in keys if d is not None and k in d and d[k] is not None}


ROOFLINE_KEYS = ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'kernel', 'kernel_avg_us', 'launches_timed', 'chain_avg_us',
                 'score_decode_avg_us', 'frac_all_l2', 'model_falsified', 'frac_split', 'peak_split', 'hbm_frac_measured',
                 'algorithmic_bytes_per_launch', 'algorithmic_flops_per_launch', 'step_latency_us', 'traffic_head')


def slim_roofline(rf):
    out = _pick(rf, ROOFLINE_KEYS)
    out.setdefault('traffic', None)
    if len(str(out.get('kernel', ''))) > 64:
        out['kernel'] = out['kernel'][:61] + '...'
    if isinstance(rf.get('l1_pipeline'), dict):
        out['l1_pipeline'] = _pick(rf['l1_pipeline'], ('floor_us', 'frac'))
    return out


def other_config_line(r):
    """One `{"other_config": label, ...}` line per side measurement, printed BEFORE the result line (every number of it is
    in gpurun_out/bench_full.json too)."""
    if 'error' in r:
        return json.dumps({'other_config': r.get('workload'), 'error': str(r['error'])[:300]})
    rf, par = r.get('roofline', {}), r.get('parity', {})
    # (value in tokens/s like the result line's; the ceiling's `peak` and units are in gpurun_out/bench_full.json)
    d = {'other_config': r.get('workload'), 'value': r.get('value'), 'ms_per_step': r.get('ms_per_step'),
         'steps': r.get('steps'), 'valid_tokens_per_step': r.get('config', {}).get('valid_tokens_per_step'),
         'roofline': _pick(slim_roofline(rf), ('bound', 'frac', 'achieved', 'kernel', 'chain_avg_us', 'score_decode_avg_us',
                                               'frac_all_l2', 'model_falsified', 'frac_split', 'step_latency_us')),
         'parity': _pick(par, ('tags_equal', 'tags_compared', 'sequences_checked', 'max_score_err'))}
    line = json.dumps(_sig(d, 5))
    if len(line) > OTHER_LINE_CAP:
        d['roofline'] = _pick(d['roofline'], ('bound', 'frac', 'frac_all_l2', 'model_falsified', 'frac_split', 'chain_avg_us', 'score_decode_avg_us'))
        line = json.dumps(_sig(d, 5))
    return line


def result_line(out):
    """The ONE JSON line of the contract, <= LINE_CAP bytes: the contract's keys, `roofline` and `cpu_baseline` as numbers (the
    prose about them lives in DESIGN.md section 6), one-number summaries of the side measurements.  Everything that was measured
    goes to gpurun_out/bench_full.json (and, for the other configs, to their own earlier lines)."""
    d = _pick(out, ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling'))
    d['vs_baseline'] = out.get('vs_baseline')
    d.update(_pick(out, ('dtype', 'data')))
    cfg = dict(out.get('config', {}))
    if len(str(cfg.get('workload', ''))) > 200:
        cfg['workload'] = cfg['workload'][:197] + '...'
    d['config'] = cfg
    d['roofline'] = slim_roofline(out.get('roofline', {}))
    cb = out.get('cpu_baseline')
    if cb:
        d['cpu_baseline'] = _pick(cb, ('value', 'unit', 'cores', 'kind', 'host_threads', 'cpu_quota_cores'))
        d['cpu_baseline']['sample'] = str(cb.get('sample', ''))[:160]
    if out.get('cpu_baseline_faithful'):
        d['cpu_baseline_faithful'] = _pick(out['cpu_baseline_faithful'], ('value', 'unit', 'cores', 'kind'))
    if out.get('parity'):
        d['parity'] = _pick(out['parity'], ('tags_equal', 'tags_compared', 'sequences_checked', 'max_score_err', 'batch_checked'))
    if out.get('gather'):
        d['gather'] = {k: (str(v)[:96] if isinstance(v, str) else v) for k, v in out['gather'].items()}
    optional = []
    if out.get('compact'):
        d['compact'] = _pick(out['compact'], ('value', 'ms_per_step', 'kernel_avg_us', 'bytes_per_token', 'tags_equal_dense'))
        optional.append('compact')
    if out.get('pipelined'):
        d['pipelined'] = _pick(out['pipelined'], ('streams', 'value', 'ms_per_step'))
        optional.append('pipelined')
    if out.get('host_inclusive'):
        d['host_inclusive'] = _pick(out['host_inclusive'], ('value', 'us_per_batch', 'same_predictions'))
        optional.append('host_inclusive')
    if out.get('other_configs'):
        # label -> ms per step (the full result of each is its own earlier line); errors by name
        d['other_configs_ms_per_step'] = {o.get('workload'): (o.get('ms_per_step') if 'error' not in o else 'error') for o in out['other_configs']}
        d['other_configs_parity'] = all(o.get('parity', {}).get('tags_equal') is True for o in out['other_configs'] if 'error' not in o) \
            and not any('error' in o for o in out['other_configs'])
        optional.append('other_configs_ms_per_step')
    if out.get('full'):
        d['full'] = out['full']
    line = json.dumps(_sig(d))
    for k in ['cpu_baseline_faithful'] + optional[::-1]:       # never needed at today's sizes; the cap holds whatever is added later
        if len(line) <= LINE_CAP:
            break
        d.pop(k, None)
        line = json.dumps(_sig(d))
    if len(line) > LINE_CAP:
        raise AssertionError('bench.py: result line of {} bytes exceeds the {} byte cap'.format(len(line), LINE_CAP))
    return line


def write_full(out):
    """the whole measurement (every note, every side result) beside the line: gpurun_out/ is merged back from the GPU box"""
    try:
        path = os.path.join(ROOT, FULL_PATH)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, 'w') as f:
            json.dump(out, f, indent=1)
        return FULL_PATH
    except OSError:
        return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--workload', default='ifst', choices=sorted(WORKLOADS))
    ap.add_argument('--batch', type=int, default=256, help='sequences per GPU')
    ap.add_argument('--states', type=int, default=0,
                    help='automaton states S (decomp: default 104; 134 = the shipped configurations with --additional_states 30; '
                         'ifst / ifst_crf: default 71; 104 = the reference\'s SNIPS-BIO / ATIS-ZH-BIO automata)')
    ap.add_argument('--vocab', type=int, default=0,
                    help='dry runs only: override the workload\'s vocabulary size (the line then says so; never a headline number)')
    ap.add_argument('--seqlen', type=int, default=64)
    ap.add_argument('--rank', type=int, default=50, help='decomp: CP rank')
    ap.add_argument('--farnn', type=int, default=0, help='decomp: gate mode 0/1/2 (reference --farnn)')
    ap.add_argument('--semiring', default='sum', choices=['sum', 'max'], help='decomp: reference --train_mode')
    ap.add_argument('--full-length', action='store_true', help='all sequences at full length')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=10.0)
    ap.add_argument('--crf', action='store_true',
                    help='decomp: CRF-Viterbi decode (reference --use_crf 1: what every shipped example configuration uses); '
                         'train: CRF negative log-likelihood instead of the cross-entropy')
    ap.add_argument('--gather', default='torch', choices=['torch', 'native'],
                    help='N > 1: the tag gather through torch.distributed (backend "nccl" = RCCL) or through the torch-free C-ABI '
                         'collective (include/farnn_rccl.h: ncclAllGather on a side HIP stream) -- the A/B of the first 8-GPU run')
    ap.add_argument('--batches', type=int, default=4,
                    help='different batches (same lengths, fresh tokens) the timed region rotates through; 1 = replay one batch')
    ap.add_argument('--streams', type=int, default=1,
                    help='in-flight batches for the main timed region: steps alternate over this many '
                         'HIP streams, each with its own model handle and workspace')
    ap.add_argument('--no-pipelined', action='store_true', help='skip the extra 2-streams measurement')
    ap.add_argument('--no-other-configs', action='store_true',
                    help='default workload only: skip the ifst_crf / decomp / fst4 side measurements')
    ap.add_argument('--no-parity', action='store_true', help='skip the post-run oracle check')
    ap.add_argument('--graph', type=int, default=0,
                    help='N > 0: capture N consecutive steps into one HIP graph and replay it (steps must be a '
                         'multiple of N); 0 = plain stream launches')
    ap.add_argument('--event-stride', type=int, default=-1,
                    help='time the kernels of every N-th step with HIP events (0 = never; default: chosen from '
                         '--steps so that at least 8 launches are timed, at most every 16th step)')
    a = ap.parse_args()
    if a.states > 0:
        d_, v_, s_, c_ = WORKLOADS[a.workload]
        WORKLOADS[a.workload] = (d_ + ' [S = {}]'.format(a.states), v_, a.states, c_)
    if a.vocab > 0:
        d_, _, s_, c_ = WORKLOADS[a.workload]
        WORKLOADS[a.workload] = (d_ + ' [DRY RUN: vocabulary reduced to {}]'.format(a.vocab), a.vocab, s_, c_)
    return a


def auto_event_stride(steps, name=None):
    """>= 8 timed launches whenever steps >= 8; never denser than needed (an event pair costs launch latency).  The
    millisecond-scale steps (fst4, synth512) time EVERY launch: a pair's ~6 us are nothing there, and a sampled average over 9
    of 60 launches read 2.7 % above the region's own mean (round 4's `fst4` line: kernel_avg_us > ms_per_step)."""
    if name in ('fst4', 'synth512'):
        return 1
    return max(1, min(16, steps // 8))


# ------------------------------------------------------------------------------------------ N > 1 launcher
def launch_ranks(a):
    """`bench.py --gpus N` without an external torchrun: start the N ranks from this process, which has not
    touched the GPU (no exec of a GPU-initialised process), relay rank 0's JSON line, exit with their status."""
    import socket
    one_dev = os.environ.get('FARNN_BENCH_ONE_DEVICE') == '1'
    if not one_dev:
        have = torch.cuda.device_count()          # counts devices without initialising the GPU
        if have < a.gpus:
            raise SystemExit('bench.py --gpus {}: only {} GPU(s) visible (FARNN_BENCH_ONE_DEVICE=1 runs every rank '
                             'on cuda:0 over gloo, for tests)'.format(a.gpus, have))
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(a.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('OMP_NUM_THREADS', '8')
    p = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith('{"metric"'):
            line = ln
        elif ln.strip():
            print(ln, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    elif p.returncode == 0:
        raise SystemExit('bench.py: the ranks exited cleanly but printed no result line')
    sys.exit(p.returncode)


# ------------------------------------------------------------------------------------------ workloads
def build_workload(name, B, L, rank_id, cp_rank, full_length, farnn=0, semiring='sum', crf=False):
    """Returns (handle, x, lengths, extras).  Weights use one seed on every rank (replicated
    model); the batch is seeded per rank (each rank owns a different shard).  `extras` keeps the host
    copies the post-run oracle check needs."""
    from re2nn_seq_amd import _lib, synth
    _, V, S, C = WORKLOADS[name]
    wrng = np.random.RandomState(1234)
    brng = np.random.RandomState(4321 + rank_id)
    dev = torch.cuda.current_device()
    extras = {}
    if name == 'synth512':
        # generated on the device (21 GB would take minutes on the host): automaton-like 0/1 tensor,
        # ~2 successors per (word, from-state) row on average, zero pad row
        g = torch.Generator(device='cuda'); g.manual_seed(1234)
        dv = torch.device('cuda', dev)
        T = torch.empty((V, S, S), dtype=torch.float32, device=dv)
        for v0 in range(0, V, 500):
            T[v0:v0 + 500] = (torch.rand((min(500, V - v0), S, S), device=dv, generator=g) < 2.0 / S).float()
        T[V - 1] = 0
        W = torch.zeros((S, S), device=dv); W[0, 0] = 1; W[S - 1, S - 1] = 1
        O = torch.zeros((C, S), device=dv)
        O[torch.randint(0, C - 1, (S,), device=dv, generator=g), torch.arange(S, device=dv)] = 1
        O[:, 0] = 0; O[:, S - 1] = 0; O[C - 1, 0] = 1; O[C - 1, S - 1] = 1
        h0 = torch.zeros(S, device=dv); h0[0] = 1
        hT = torch.zeros(S, device=dv); hT[0] = 1; hT[S - 1] = 1
        h = _lib.create_onehot_ifst(T, W, O, h0, hT, nl='tanh', device=dev)   # tanh keeps states bounded
        # (the generated tensor stays on the device -- 21 GB of 288 -- for the post-run check: the blocks a few sequences touch
        #  are read back and the numpy oracle walks those sequences)
        extras.update(T_dev=T, W=W.cpu().numpy(), O=O.cpu().numpy(), h0=h0.cpu().numpy(), hT=hT.cpu().numpy())
        del T
    elif name in ('ifst', 'ifst_crf'):
        T, W, O, h0, hT = synth.random_ifst_tensors(V, S, C, wrng)
        crf = name == 'ifst_crf'
        tr = None
        if crf:
            K = C + 2
            tr = (wrng.randn(K, K) * 0.1).astype(np.float32)
            tr[:, K - 2] = -10000.0
            tr[K - 1, :] = -10000.0
        h = _lib.create_onehot_ifst(T, W, O, h0, hT, use_crf=crf, crf_trans=tr, device=dev)
        extras.update(T=T, W=W, Tf=T + W, O=O, h0=h0, hT=hT, tr=tr)
    elif name == 'fst4':
        T, W, O, h0, hT = synth.random_ifst_tensors(V, S, C, wrng)
        # the 4-D layout of the same automaton: label c lives on the edges into states with O[c,j]=1
        T4 = np.einsum('vsj,cj->vcsj', T, O).astype(np.float32)
        W4 = np.einsum('sj,cj->csj', W, O).astype(np.float32)
        h = _lib.create_onehot_fst4(T4, W4, h0, hT, device=dev)
        extras.update(T4=T4, W4=W4, h0=h0, hT=hT)
    elif name in ('decomp1', 'decomp0'):
        p = synth.random_decomposed_params(V, S, C, cp_rank, 100, wrng, contractive=True)
        RO = 70
        f = lambda *shape, sc=0.2: (wrng.randn(*shape) * sc).astype(np.float32)      # noqa: E731
        q = {'Vgen': p['V_embed'].astype(np.float32), 'S1': p['S1'].astype(np.float32),
             'S2': p['S2'].astype(np.float32), 'h0': p['start_vector'].astype(np.float32),
             'hT': p['final_vector'].astype(np.float32), 'farnn': 0, 'nl': 2, 'semiring': 1 if semiring == 'max' else 0,
             'sig_k': 5}
        if name == 'decomp1':
            # the scoring GEMM of one token: U = abw[S,S] . S2o[S,RO], the a b~ scaling and br = sum S1o . U
            extras['mfma_flops_per_token'] = 2.0 * S * S * RO + 2.0 * S * S + 2.0 * S * RO
            q.update(W=p['wildcard_mat'].astype(np.float32), Cout=f(C, RO, sc=0.5), S1o=f(S, RO), S2o=f(S, RO))
            h = _lib.create_decomp_ind1(q['Vgen'], q['S1'], q['S2'], q['W'], q['Cout'], q['S1o'], q['S2o'],
                                        q['h0'], q['hT'], nl='tanh', semiring=semiring, device=dev)
        else:
            # label factors whose column sums stay of order one (the recurrence sees Vgen * sum_c C and
            # sum_q (sum_c Cw) S1w S2w^T + WW): a well-conditioned model, like `decomp` (synth.random_decomposed_params)
            Cm = wrng.randn(C, cp_rank) * 0.5
            Cm = Cm - Cm.mean(0) + 1.0 / C                        # entries of order one, every column sums to exactly 1
            q.update(C=Cm.astype(np.float32), Cw=f(C, RO, sc=0.3 / np.sqrt(C)),
                     S1w=f(S, RO, sc=0.2 / np.sqrt(RO)), S2w=f(S, RO, sc=0.2 / np.sqrt(RO)),
                     WW=p['wildcard_mat'].astype(np.float32))
            h = _lib.create_decomp_fst(q['Vgen'], q['C'], q['S1'], q['S2'], q['Cw'], q['S1w'], q['S2w'], q['WW'],
                                       q['h0'], q['hT'], nl='tanh', semiring=semiring, device=dev)
        extras['q'] = q
    else:
        # synth.snips_sized_model: the generator this branch always used (seed 1234, beta = 1: the generalized table is V_embed
        # itself), plus -- with `crf` -- the START / STOP rows of the output matrix and the transitions of a CRF over the C
        # labels (model_decompose_single.py:78-79, crf.py:31-46): `--rank 250 --farnn 2 --crf` is the configuration every
        # shipped model_seq/example/*.res was trained with
        _, q, gates, tr = synth.snips_sized_model(cp_rank, farnn, crf, seed=1234, S=S, V=V, C=C)
        q['semiring'] = 1 if semiring == 'max' else 0
        h = _lib.create_decomp_ifst(q['Vgen'], q['S1'], q['S2'], q['W'], q['Cout'], q['h0'], q['hT'], nl='tanh', farnn=farnn,
                                    gates=gates, sigmoid_exponent=5, semiring=semiring, use_crf=crf, crf_trans=tr, device=dev)
        extras['q'] = q
        extras['tr'] = tr
    x, lengths = synth.random_batch(V, B, L, brng)
    if full_length:
        lengths[:] = L
        x, _ = synth.random_batch(V, B, L, brng, min_len=L)
    order = os.environ.get('FARNN_BENCH_ORDER')
    if order:        # experiment: input permutations that change which sequences share a CU
        idx = np.argsort(-lengths, kind='stable')
        if order == 'fold':
            half = B // 2
            idx = np.concatenate([idx[:half], idx[half:][::-1]])
        x, lengths = x[idx], lengths[idx]
    return h, x, lengths, extras


# ------------------------------------------------------------------------------------------ parity (after the timed region)
def parity_check(name, h, extras, x, lengths, gpu_tags, dev):
    """Oracle check of the batch that was timed: the tags the LAST timed step left in HBM (`gpu_tags`,
    int32 [B,L], -1 at pads) and, for the floating-point paths, a fresh call that also returns scores.
    Heavy oracles (4-D gather, independent=0/1 scoring) check the first few sequences only; sequences
    are independent, so a prefix of the batch is a valid sub-batch."""
    from oracle import farnn_oracle as fo
    from re2nn_seq_amd import _lib
    B, L = x.shape
    mask = np.arange(L)[None, :] < lengths[:, None]
    out = {'tags_equal': None, 'sequences_checked': B, 'max_score_err': None}

    def gpu_scores(n):
        xs = torch.from_numpy(np.ascontiguousarray(x[:n])).to(dev)
        ls = torch.from_numpy(np.ascontiguousarray(lengths[:n])).to(dev)
        K = h.num_columns()
        sc = torch.empty((n, L, K), dtype=torch.float32, device=dev)
        tg = torch.empty((n, L), dtype=torch.int32, device=dev)
        h.tag(xs.data_ptr(), ls.data_ptr(), n, L, _lib.MODE_LOCAL, tg.data_ptr(), None, sc.data_ptr())
        torch.cuda.synchronize(dev)
        return sc.cpu().numpy(), tg.cpu().numpy()

    def float_compare(ref, n, crf_tr=None):
        """ref [n, Lmax, K] oracle scores: scores to 1e-4, tags equal wherever the oracle's decision margin
        exceeds 2e-4, twice the score bar (argmax paths) / everywhere (the check the tests apply at this size)."""
        sc, tg = gpu_scores(n)
        Lmax = ref.shape[1]
        m = mask[:n, :Lmax]
        err = float(np.abs(sc[:, :Lmax][m] - ref[m]).max())
        out['max_score_err'] = err
        out['sequences_checked'] = n
        rt = fo.decode_argmax(ref, 0.5, 0)
        refc = ref.copy(); refc[..., -1] = np.minimum(refc[..., -1], 0.5)
        top2 = np.sort(refc[m], axis=1)[:, -2:]
        safe = (top2[:, 1] - top2[:, 0]) > 2e-4
        same_fresh = np.array_equal(tg[:, :Lmax][m][safe], rt[m][safe])
        same_timed = np.array_equal(gpu_tags[:n, :Lmax][m][safe], rt[m][safe])
        out['tags_equal'] = bool(same_fresh and same_timed and err <= 1e-4 * max(1.0, float(np.abs(ref[m]).max())))
        out['tags_compared'] = int(safe.sum())
        out['tags_within_margin_skipped'] = int((~safe).sum())

    if name in ('ifst', 'ifst_crf'):
        from oracle import c_port
        tags, scores, _ = c_port.onehot_ifst_tag(extras['Tf'], extras['O'], extras['h0'], extras['hT'], x, lengths,
                                                 want_scores=(name == 'ifst_crf'), nthreads=min(os.cpu_count() or 1, 16))
        if name == 'ifst':
            out['tags_equal'] = bool(np.array_equal(tags, gpu_tags))
            out['tags_compared'] = int(tags.size)
            out['tags_within_margin_skipped'] = 0
            out['oracle'] = 'C port of the oracle, every position of every sequence, bit-exact'
        else:
            ext = fo.onehot_crf_extension_scores(scores)
            ref = fo.decode_crf(ext, lengths, extras['tr'], 0.5, 0)
            out['tags_equal'] = bool(np.array_equal(ref[mask], gpu_tags.astype(np.int64)[mask]))
            out['tags_compared'] = int(mask.sum())
            out['tags_within_margin_skipped'] = 0
            out['oracle'] = 'C-port scores + numpy Viterbi (crf.py:102-195), every valid position, bit-exact'
    elif name == 'decomp' and extras.get('tr') is not None:
        # CRF decode (model_decompose.py:349-356): scores to 1e-4 against the oracle; the Viterbi DP is the same f32 expression
        # on both sides, so the oracle's decode of the GPU's OWN scores must equal the tags bit for bit -- the timed launch's
        # (the fused score + Viterbi kernel: no score tensor) and a fresh call's (scores written out)
        ref = fo.decomp_ifst_scores(extras['q'], x, lengths)
        sc, tg = gpu_scores(B)
        Lmax = ref.shape[1]
        m = mask[:, :Lmax]
        err = float(np.abs(sc[:, :Lmax][m] - ref[m]).max())
        own = fo.decode_crf(sc[:, :Lmax], lengths, extras['tr'], 0.5, 0)
        out['max_score_err'] = err
        out['tags_equal'] = bool(np.array_equal(own[m], tg[:, :Lmax][m].astype(np.int64)) and
                                 np.array_equal(own[m], gpu_tags[:, :Lmax][m].astype(np.int64)) and
                                 err <= 1e-4 * max(1.0, float(np.abs(ref[m]).max())))
        out['tags_compared'] = int(m.sum())
        out['tags_within_margin_skipped'] = 0
        out['oracle'] = ('numpy oracle: decomp_ifst_scores <= 1e-4 on the whole batch; the oracle\'s Viterbi (crf.py:102-195) on the '
                         'GPU\'s own scores equals the timed tags at every valid position, bit-exact')
    elif name == 'decomp':
        ref = fo.decomp_ifst_scores(extras['q'], x, lengths)
        float_compare(ref, B)
        out['oracle'] = 'numpy oracle decomp_ifst_scores, whole batch: scores <= 1e-4, tags outside 2e-4 margins'
    elif name in ('decomp1', 'decomp0'):
        n = min(B, 8)
        fn = fo.decomp_ind1_scores if name == 'decomp1' else fo.decomp_fst_scores
        ref = fn(extras['q'], x[:n], lengths[:n])
        float_compare(ref, n)
        out['oracle'] = 'numpy oracle, first {} sequences: scores <= 1e-4, tags outside 2e-4 margins'.format(n)
    elif name == 'fst4':
        n = min(B, 32)
        ref = fo.onehot_fst4_scores(extras['T4'], extras['W4'], extras['h0'], extras['hT'], x[:n], lengths[:n])
        rt = fo.decode_argmax(ref, 0.5, 0)
        out['tags_equal'] = bool(np.array_equal(rt[mask[:n]], gpu_tags[:n].astype(np.int64)[mask[:n]]))
        out['sequences_checked'] = n
        out['oracle'] = 'numpy oracle onehot_fst4_scores, first {} sequences, bit-exact tags'.format(n)
    elif name == 'synth512' and 'T_dev' in extras:
        # the longest, the shortest and four more sequences: the blocks they touch come back from the device (the tensor was
        # generated there), the numpy oracle walks them -- scores to 1e-4, tags outside 2e-4 margins (tanh between the steps)
        rows = np.unique(np.concatenate([[int(np.argmax(lengths)), int(np.argmin(lengths))], np.linspace(0, B - 1, 4).astype(int)]))
        xs, ls = x[rows].copy(), lengths[rows].copy()
        toks, inv = np.unique(xs, return_inverse=True)
        Tsub = extras['T_dev'][torch.from_numpy(toks).to(extras['T_dev'].device)].cpu().numpy()
        xc = inv.reshape(xs.shape).astype(np.int64)
        ref = fo.onehot_ifst_scores(Tsub, extras['W'], extras['O'], extras['h0'], extras['hT'], xc, ls, nl=fo.NL_TANH)
        n = len(rows)
        xd_ = torch.from_numpy(np.ascontiguousarray(xs)).to(dev); ld_ = torch.from_numpy(np.ascontiguousarray(ls)).to(dev)
        K = h.num_columns()
        sc = torch.empty((n, L, K), dtype=torch.float32, device=dev)
        tg = torch.empty((n, L), dtype=torch.int32, device=dev)
        h.tag(xd_.data_ptr(), ld_.data_ptr(), n, L, _lib.MODE_LOCAL, tg.data_ptr(), None, sc.data_ptr())
        torch.cuda.synchronize(dev)
        Lmax = ref.shape[1]
        m = (np.arange(Lmax)[None, :] < ls[:, None])
        err = float(np.abs(sc.cpu().numpy()[:, :Lmax][m] - ref[m]).max())
        rt = fo.decode_argmax(ref, 0.5, 0)
        refc = ref.copy(); refc[..., -1] = np.minimum(refc[..., -1], 0.5)
        top2 = np.sort(refc[m], axis=1)[:, -2:]
        safe = (top2[:, 1] - top2[:, 0]) > 2e-4
        same_fresh = np.array_equal(tg.cpu().numpy()[:, :Lmax][m][safe], rt[m][safe])
        same_timed = np.array_equal(gpu_tags[rows][:, :Lmax][m][safe], rt[m][safe])
        out.update(max_score_err=err, sequences_checked=n, tags_compared=int(safe.sum()), tags_within_margin_skipped=int((~safe).sum()),
                   tags_equal=bool(same_fresh and same_timed and err <= 1e-4 * max(1.0, float(np.abs(ref[m]).max()))),
                   oracle='numpy oracle on the blocks read back from the device, {} whole sequences (longest, shortest, four more): '
                          'scores <= 1e-4, tags outside 2e-4 margins'.format(n))
    else:
        out['oracle'] = 'none at this size (weights exist on the device only); covered by tests/test_gpu_fullsize_properties.py'
        out['sequences_checked'] = 0
    return out


# ------------------------------------------------------------------------------------------ CPU baselines
def cpu_baseline(extras, x, lengths, seconds):
    """The C port of the oracle on ALL of this host's cores (north_star: "all host cores, count stated"), same batch, bounded
    sample: the "fair" CPU number (T+W hoisted, stops at len).  Round 5: the barrier-free throughput form
    (oracle_onehot_ifst_tag_stream: passes x B whole sequences -- both chains, the score rows, the decode -- dealt to the threads
    dynamically); the two-phase form (chains, barrier, score rows: round 4) peaked at 16-32 of 256 threads and is timed beside it.
    A short sweep picks the fastest thread count; `cores` reports the threads used, `cores_all` the rate on every hardware thread."""
    from oracle import c_port
    c_port.load(native=True)
    ncpu = os.cpu_count() or 1
    args = (extras['Tf'], extras['O'], extras['h0'], extras['hT'], x, lengths)
    tok = int(lengths.sum())

    def rate(nthreads, budget, stream=True):
        c_port.onehot_ifst_tag(*args, nthreads=nthreads, reps=2, stream=stream)
        chunk = 20
        n, t0 = 0, time.perf_counter()
        while True:
            c_port.onehot_ifst_tag(*args, nthreads=nthreads, reps=chunk, stream=stream)
            n += chunk
            el = time.perf_counter() - t0
            if el >= budget or n >= 40000:
                return tok * n / el, n, el
            if el < 0.1 * budget:
                chunk = min(chunk * 2, 1280)

    # what this process may actually use: the container's CPU-time quota (cgroup v2 cpu.max = "quota period"; the GPU boxes of this
    # pool run every job under 16 cores' worth of a 256-thread host -- more threads than that only get throttled)
    quota = None
    try:
        q_, p_ = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if q_ != 'max':
            quota = float(q_) / float(p_)
    except (OSError, ValueError):
        pass
    cands = sorted({c for c in (ncpu, ncpu // 2, ncpu // 4, 64, 32, 16, 8, int(quota) if quota else 0) if 1 <= c <= ncpu})
    sweep = {c: rate(c, 0.5)[0] for c in cands}
    best = max(cands, key=lambda c: sweep[c])
    value, n, el = rate(best, seconds)
    two_phase = max(rate(c, 0.4, stream=False)[0] for c in sorted({c for c in (16, 32, 64) if c <= ncpu} or {ncpu}))
    # (the tags of both forms equal the numpy oracle's and the reference's: tests/test_oracle_c.py)
    return {'value': value, 'unit': 'tokens/s', 'cores': int(best), 'host_threads': ncpu, 'cpu_quota_cores': quota, 'kind': 'port',
            'rate_by_threads': {str(c): round(v) for c, v in sweep.items()}, 'two_phase_form_best': two_phase,
            'sample': '{} passes of the same {}x{} batch ({} valid tokens) in {:.1f} s; C port of the '
                      'oracle (T+W hoisted; OpenMP over passes x sequences, longest first, no barrier: one parallel region, each '
                      'thread walks both chains of a sequence and scores it), best of thread counts {} '
                      'on a {}-thread host{}'.format(n, x.shape[0], x.shape[1], tok, el, cands, ncpu,
                                                     ' whose container grants {:g} cores of CPU time (cgroup cpu.max)'.format(quota) if quota else '')}


def cpu_baseline_faithful(extras, x, lengths, seconds):
    """The reference's own algorithm (model_onehot.py:366-403) as the numpy oracle restates it: re-adds T+W on
    every call, gathers B x S x S blocks per step, walks all L padded steps, batched einsum on the host BLAS
    threads -- the "reference-faithful" CPU number of BASELINE.md section 3 (the PyTorch reference itself
    cannot travel to the GPU box)."""
    from oracle import farnn_oracle as fo
    tok = int(lengths.sum())
    args = (extras['T'], extras['W'], extras['O'], extras['h0'], extras['hT'], x, lengths)
    fo.decode_argmax(fo.onehot_ifst_scores(*args), 0.5, 0)
    n, t0 = 0, time.perf_counter()
    while True:
        fo.decode_argmax(fo.onehot_ifst_scores(*args), 0.5, 0)
        n += 1
        el = time.perf_counter() - t0
        if el + el / n > seconds:
            break
    return {'value': tok * n / el, 'unit': 'tokens/s', 'cores': os.cpu_count() or 1, 'kind': 'port',
            'sample': '{} passes of the same batch in {:.1f} s; numpy float32 restatement of the reference algorithm '
                      '(T+W re-added per call, all L steps, B x S x S gathers; numpy/BLAS threads of a {}-thread '
                      'host)'.format(n, el, os.cpu_count() or 1)}


def host_inclusive(extras, x, lengths, seconds=0.4):
    """The boundary's own calling convention (reference val.py:17-31): CPU tensors in, CPU tensors out through the
    model mirror FARNN_S_O_I_S -- H2D copies of x / lengths, the tagging launch, the D2H copy of the flat predictions and
    the label flatten, per batch.  Two forms: one synchronous forward_local() per batch, and the eval loop's form with
    two batches in flight (submit_local / result).  PCIe-inclusive, so never `value`."""
    import argparse as ap
    from re2nn_seq_amd.farnn.model_onehot import FARNN_S_O_I_S
    a = ap.Namespace(rand_constant=0.0, threshold=0.5, train_mode='sum', local_loss_func='CE1', use_priority=0,
                     independent=2, update_nonlinear='none')
    S = extras['h0'].shape[0]
    m = FARNN_S_O_I_S(extras['T'], extras['O'], extras['W'], np.zeros(S), extras['hT'], extras['h0'], None, a, o_idx=0)
    xt, lt = torch.from_numpy(x), torch.from_numpy(lengths)
    lab = torch.zeros_like(xt)
    tok = int(lengths.sum())
    for _ in range(20):
        m.forward_local(xt, lab, lt, train=False)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(50):
            m.forward_local(xt, lab, lt, train=False)
        n += 50
    sync_el = time.perf_counter() - t0
    import collections
    q, n2, t0 = collections.deque(), 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(50):
            q.append(m.submit_local(xt, lab, lt))
            if len(q) > m.pipeline_depth:
                q.popleft().result()
        n2 += 50
    while q:
        last = q.popleft().result()
    pipe_el = time.perf_counter() - t0
    ok = bool(np.array_equal(last[1].numpy(), m.forward_local(xt, lab, lt, train=False)[1].numpy()))
    m.invalidate()
    return {'value': tok * n2 / pipe_el, 'unit': 'tokens/s', 'us_per_batch': pipe_el / n2 * 1e6,
            'batches_in_flight': m.pipeline_depth,
            'synchronous': {'value': tok * n / sync_el, 'us_per_batch': sync_el / n * 1e6},
            'same_predictions': ok,
            'note': 'CPU tensors in / out through FARNN_S_O_I_S.forward_local (farnn_tag_host_submit / _wait: mapped pinned staging, a staging kernel, the tagging launch storing its flat predictions into mapped host memory); PCIe-inclusive, '
                    'never the headline value'}


def batch_variants(name, x, lengths, rank_id, n=4):
    """n batches for the timed region to rotate through: the workload's own batch and n - 1 more with the SAME lengths (so the
    tokens per step, `value` and `ms_per_step` mean what they meant) but tokens drawn afresh -- a different Zipf-rank -> word
    permutation each, i.e. different hot words: the L2 hit share of the block gather is not that of one replayed batch."""
    from re2nn_seq_amd import synth
    _, V, _, _ = WORKLOADS[name]
    B, L = x.shape
    out = [x]
    for k in range(1, n):
        rng = np.random.RandomState(4321 + rank_id + 7919 * k)
        xk, _ = synth.random_batch(V, B, L, rng, min_len=L, full_length_rows=B)
        xk[np.arange(L)[None, :] >= lengths[:, None]] = V - 1
        out.append(xk)
    return out


# ------------------------------------------------------------------------------------------ roofline
def load_traffic(name, a, kname=''):
    """Measured fabric-side bytes per launch of the dominant kernel (profiles/traffic.json, separate PMC passes),
    valid only for the shape -- and the form of the kernel: the one-launch step has an entry of its own -- it was measured on."""
    shape_ok = {'ifst': (256, 64), 'fst4': (256, 64), 'synth512': (1024, 128)}.get(name)
    if shape_ok != (a.batch, a.seqlen) or a.full_length:
        return None
    S = WORKLOADS[name][2]
    if S != BASE_STATES[name]:
        name = '{}_s{}'.format(name, S)              # a split measured at another state count has its own entry
    elif name == 'ifst' and 'fused' in kname:
        name = 'ifst_one_launch'
    tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
    if not os.path.exists(tpath):
        return None
    with open(tpath) as f:
        return json.load(f).get(name, {}).get('hbm_bytes_per_launch')


def traffic_head():
    """which commit's kernels profiles/traffic.json was measured on"""
    tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
    try:
        with open(tpath) as f:
            return json.load(f).get('_measured_at', {}).get('head')
    except (OSError, ValueError):
        return None


def roofline_for(name, a, h, extras, tok_local, chain_avg_s, score_avg_s, n_timed):
    """The dominant kernel (by measured time) against the ceiling that bounds it.
      dense-block recurrence / 4-D scoring stream : bytes.  Working set above the 256 MiB Infinity Cache ->
          HBM (8.0 TB/s spec).  Below it the blocks are served by L2 and the Infinity Cache and never by HBM:
          the ceiling is the guide's row-gather rates, split by the measured fabric traffic (bytes that left L2
          at the 8.6 TB/s Infinity-Cache gather rate, the rest at the 16.8 TB/s L2 gather rate); the HBM-side
          rate actually measured is reported beside it as hbm_frac_measured.
      decomposed recurrences, Viterbi : shared weights in LDS, ~R*4 bytes per token from HBM -> not a bandwidth
          problem by construction; priced in f32 VALU flops against the 157.3 TFLOP/s vector
          peak, with the per-step latency (the real bound of a 64-step serial chain) beside it.
      independent=1 scoring : f32 MFMA."""
    from re2nn_seq_amd import _lib
    _, V, S, C = WORKLOADS[name]
    dom = _lib.KERN_SCORE if score_avg_s > chain_avg_s else _lib.KERN_CHAIN
    dom_s = score_avg_s if dom == _lib.KERN_SCORE else chain_avg_s
    kname = h.kernel_name(dom)
    rf = {'kernel': kname, 'kernel_avg_us': dom_s * 1e6, 'launches_timed': n_timed,
          'chain_avg_us': chain_avg_s * 1e6, 'score_decode_avg_us': score_avg_s * 1e6, 'traffic': None}
    if dom_s <= 0:
        rf.update(bound='hbm', achieved=0.0, peak=HBM_PEAK_GBS, unit='GB/s', frac=0.0)
        return rf
    L = a.seqlen
    byte_bound = (('chain_kernel' in kname or 'chain_regs_kernel' in kname or 'chain_wide_kernel' in kname) and dom == _lib.KERN_CHAIN) or 'fst4' in kname
    if byte_bound:
        alg = h.kernel_algorithmic_bytes(dom, tok_local)
        achieved = alg / dom_s / 1e9
        traffic = load_traffic(name, a, kname)
        SP = (S + 3) // 4 * 4
        ws = (2.0 * V * S * SP * 4) if 'chain' in kname else (1.0 * V * C * S * SP * 4)
        rf.update(achieved=achieved, unit='GB/s', traffic=traffic, algorithmic_bytes_per_launch=alg,
                  working_set_bytes=ws, traffic_head=traffic_head() if traffic is not None else None)
        if ws > INFINITY_CACHE_BYTES:
            peak = HBM_PEAK_GBS
            if traffic is not None and 0 < traffic < alg:
                # Zipf-distributed words: part of the algorithmic bytes are re-reads served by L2 / the Infinity Cache and
                # never reach HBM.  Floor = the measured HBM-side bytes at the HBM peak + the re-reads at the L2 gather rate
                # (the higher of the two cache ceilings, so the fraction is never flattered); r02a priced all of it at
                # 8 TB/s and printed 1.01.
                t_floor = traffic / (HBM_PEAK_GBS * 1e9) + (alg - traffic) / (L2_GATHER_GBS * 1e9)
                peak = alg / t_floor / 1e9
                rf['peak_note'] = ('{:.1f} GB per launch leave L2 (fabric-side counter: Infinity-Cache hits included; priced at the 8.0 TB/s '
                                   'HBM peak), {:.1f} GB are L2 hits (16.8 TB/s L2 gather rate)'.format(traffic / 1e9, (alg - traffic) / 1e9))
            rf.update(bound='hbm', peak=peak)
        else:
            # no traffic measurement for this shape: every row could be an L2 hit, so price against the L2 gather rate
            peak = L2_GATHER_GBS
            rf['peak_note'] = ('working set {:.0f} MB is L2/Infinity-Cache resident; no PMC traffic split for this shape: '
                               'priced against the L2 row-gather rate (16.8 TB/s), an upper ceiling'.format(ws / 1e6))
            if traffic is not None and 0 < traffic < alg:
                t_floor = (alg - traffic) / (L2_GATHER_GBS * 1e9) + traffic / (IC_GATHER_GBS * 1e9)
                peak = alg / t_floor / 1e9
                rf['peak_note'] = ('working set {:.0f} MB is L2/Infinity-Cache resident: {:.0f} MB per launch leave L2 '
                                   '(measured, Infinity-Cache gather 8.6 TB/s), the other {:.0f} MB are L2 hits (L2 gather '
                                   '16.8 TB/s); HBM is not on the path'.format(ws / 1e6, traffic / 1e6, (alg - traffic) / 1e6))
            # ONE rule, fixed before the run: `peak` is the modelled split (or the L2 gather rate where no
            # split was measured).  A kernel that beats the split falsifies the MODEL (the guide's rates are measured on one loop
            # shape, and fills from the Infinity Cache overlap with L2 hits here), not the hardware: the line then says
            # `model_falsified` and prices EVERY byte at the L2 gather rate -- `peak` / `frac` are that stricter reading, the split
            # stays beside them as `peak_split` / `frac_split`.  No line prints a fraction above 1.
            rf['frac_all_l2'] = achieved / L2_GATHER_GBS
            if achieved > peak:
                rf.update(model_falsified=True, peak_split=peak, frac_split=achieved / peak)
                peak = L2_GATHER_GBS
            rf.update(bound='infinity_cache', peak=peak)
        if traffic is not None:
            rf['hbm_frac_measured'] = traffic / dom_s / 1e9 / HBM_PEAK_GBS
        rf['frac'] = achieved / rf['peak']
        if 'chain_regs_kernel' in kname and S <= 72:
            # The bound that was MEASURED for this kernel at the end of round 5 (scripts/probe/ta_rate.hip,
            # scripts/debug/pool_probe.py): the compute unit's L1 pipeline -- 16 cycles per vector load instruction whatever its
            # width or active lanes, plus the 64 B/clk fill of the lines it misses, in the same pipeline.  A workgroup step is 24
            # load instructions and one block's lines; the floor spreads the launch's workgroup steps evenly over the compute units
            # at the device's MAXIMUM clock (the launch runs at ~2.2 GHz: the fraction is a lower bound).  Reported beside the
            # bandwidth reading above, which stays the line's `frac`.
            props = torch.cuda.get_device_properties(torch.cuda.current_device())
            ncu = int(getattr(props, 'multi_processor_count', 256))
            clk = float(getattr(props, 'clock_rate', 2400000)) * 1e3
            cyc = 16.0 * 24 + S * SP * 4 / 64.0
            floor_s = 2.0 * tok_local * cyc / (ncu * clk)
            rf['l1_pipeline'] = {'cycles_per_workgroup_step': cyc, 'load_instructions_per_step': 24, 'cycles_per_load_instruction': 16,
                                 'fill_bytes_per_step': S * SP * 4, 'fill_bytes_per_clock': 64, 'compute_units': ncu, 'clock_hz': clk,
                                 'floor_us': floor_s * 1e6, 'frac': floor_s / dom_s}
        return rf
    if dom == _lib.KERN_SCORE and 'mfma_flops_per_token' in extras and 'mfma' in kname:
        fl = extras['mfma_flops_per_token'] * tok_local
        ach = fl / dom_s / 1e12
        rf.update(bound='mfma', achieved=ach, peak=F32_PEAK_TFLOPS, unit='TFLOP/s', frac=ach / F32_PEAK_TFLOPS,
                  algorithmic_flops_per_launch=fl)
        return rf
    # latency-bound kernels: VALU flops of the dominant kernel per valid token
    R, K = a.rank, h.num_columns()
    if 'chain_viterbi' in kname:          # ONE launch: dense-block recurrence (both directions) + score GEMM + Viterbi DP
        fl_tok = 2.0 * (2 * S * S) + 2.0 * K * S + 3.0 * K * K
        per_step_us = dom_s * 1e6 / max(1, L)
        note = ('recurrence, scores and CRF decode of a sequence in one workgroup: three serial passes over its {} positions '
                '(chain step, Viterbi forward step, back-trace step); the blocks are L2-resident, scores and partitions stay in LDS'.format(L))
    elif dom == _lib.KERN_CHAIN:          # decomposed recurrence, both directions
        fl_tok = 2.0 * (4 * S * R + 2 * S * S)
        if a.farnn >= 1:
            fl_tok += (2.0 if a.farnn == 2 else 1.0) * (2 * S * S + 2 * S * R) * 2
        per_step_us = dom_s * 1e6 / max(1, L)
        note = ('weights shared by all sequences and LDS-resident; {:.0f} B per token from HBM: not bandwidth-bound by '
                'construction. The bound is the serial chain: {} steps per direction'.format(R * 4 + 12, L))
    else:                                 # score GEMM (+ Viterbi DP: 2 adds + 1 max per tag pair)
        fl_tok = 2.0 * K * S + (3.0 * K * K if 'viterbi' in kname else 0.0)
        per_step_us = dom_s * 1e6 / max(1, L)
        note = 'scores stay on-chip; the DP is serial over the {} positions of a sequence'.format(L)
    fl = fl_tok * tok_local
    ach = fl / dom_s / 1e12
    rf.update(bound='valu_f32', achieved=ach, peak=F32_PEAK_TFLOPS, unit='TFLOP/s', frac=ach / F32_PEAK_TFLOPS,
              algorithmic_flops_per_launch=fl, step_latency_us=per_step_us, note=note)
    return rf


def run_train(a, world, rank, dev, dist):
    """--workload train: K training steps of the decomposed i-FST (reference train_decompose.py:171-193) on
    synthetic SNIPS-sized factors: word table from the embedding bridge (torch GEMM), the library's step
    (farnn_decomp_ifst_train_step), backward through the table, Adam.  N > 1: data parallel, gradients
    all-reduced over RCCL (one flat bucket)."""
    from re2nn_seq_amd import _lib, synth
    from re2nn_seq_amd.farnn.train_step import decomp_ifst_train_step
    desc, V, S, K = WORKLOADS['train']
    if a.crf:
        K += 2                      # START / STOP tags (baselines/crf.py:31-46)
    R, D, B, L = a.rank, 100, a.batch, a.seqlen
    wrng = np.random.RandomState(1234)
    brng = np.random.RandomState(4321 + rank)

    def f(*shape, sc=0.3):
        return torch.from_numpy((wrng.randn(*shape) * sc).astype(np.float32)).to(dev).requires_grad_(True)
    Cm = np.zeros((K, S), np.float32)
    Cm[wrng.randint(0, K - (2 if a.crf else 0), size=S), np.arange(S)] = 1
    p = dict(S1=f(S, R, sc=0.1), S2=f(S, R, sc=0.1), V_embed=f(V, R, sc=0.8), G=f(D, R), E=f(V, D),
             C=torch.from_numpy(Cm).to(dev).requires_grad_(True),
             W=torch.from_numpy(((wrng.rand(S, S) < 1.0 / S) * 0.5).astype(np.float32)).to(dev).requires_grad_(True),
             h0=f(S, sc=0.5), hT=f(S, sc=0.5))
    gate_names = ('Wss1', 'Wrs1', 'bs1', 'Wss2', 'Wrs2', 'bs2')[:3 * a.farnn]
    for n in gate_names:                      # xavier-sized gates, bias as --bias_init would set it
        p[n] = f(1, S, sc=0.5) if n.startswith('bs') else (f(S, S, sc=1.0 / np.sqrt(S)) if n.startswith('Wss') else f(R, S, sc=1.0 / np.sqrt(R)))
    if a.crf:
        tr = np.zeros((K, K), np.float32)
        tr[:, K - 2] = -10000.0
        tr[K - 1, :] = -10000.0
        p['trans'] = torch.from_numpy(tr).to(dev).requires_grad_(True)
    beta = torch.full((R,), 0.7, device=dev)
    x, lengths = synth.random_batch(V, B, L, brng)
    if a.full_length:
        lengths[:] = L
    labels = brng.randint(0, K - (2 if a.crf else 0), size=(B, L)).astype(np.int64)
    xd, ld, lab = torch.from_numpy(x).to(dev), torch.from_numpy(lengths).to(dev), torch.from_numpy(labels).to(dev)
    tc = _lib.TrainContext(V, S, R, K, nl='tanh', threshold=0.5, o_idx=0, device=dev.index or 0, use_crf=a.crf,
                           farnn=a.farnn, sigmoid_exponent=5.0)
    params = list(p.values())
    ntok_local = int(lengths.sum())
    opt = torch.optim.Adam(params, lr=1e-4)

    def step():
        opt.zero_grad(set_to_none=True)
        Vgen = p['V_embed'] * beta + torch.tanh(p['E'] @ p['G']) * (1 - beta)
        loss, _ = decomp_ifst_train_step(tc, Vgen, p['S1'], p['S2'], p['W'], p['C'], p['h0'], p['hT'], None, xd, ld, lab,
                                         crf_trans=p.get('trans'), gates=tuple(p[n] for n in gate_names),
                                         valid_tokens=ntok_local)
        loss.backward()
        if world > 1:
            flat = torch.cat([q.grad.reshape(-1) for q in params])
            dist.all_reduce(flat)
            flat /= world
            o = 0
            for q in params:
                q.grad.copy_(flat[o:o + q.numel()].view_as(q))
                o += q.numel()
        opt.step()
        return loss

    for _ in range(max(a.warmup, 1)):
        step()
    tc.set_profiling(1 if a.event_stride else 0)
    tc.time()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
        n = torch.tensor([int(lengths.sum())], dtype=torch.int64, device=dev)
        dist.all_reduce(n)
        tok_total = int(n.item())
    else:
        tok_total = int(lengths.sum())
    lib_ms, lib_n = tc.time()
    if rank == 0:
        tok_local = int(lengths.sum())
        # VALU flops of the library part per valid token: forward recurrence 2(4SR+2S^2) + scoring 2KS, the
        # backward pass about twice that (BPTT products + the parameter-gradient outer products), gates on top
        fwd = 2.0 * (4 * S * R + 2 * S * S) + 2.0 * K * S
        if a.farnn:
            fwd += (2.0 if a.farnn == 2 else 1.0) * (2 * S * S + 2 * S * R) * 2
        if a.crf:
            fwd += 4.0 * K * K            # forward-backward recursions of the CRF: two dot products per tag pair
        flops = 3.0 * fwd * tok_local
        lib_s = (lib_ms / max(lib_n, 1)) * 1e-3
        achieved = flops / lib_s / 1e12 if lib_s > 0 else 0.0
        out = {
            'metric': 'trained tokens/sec @ batch=256, seqlen=64 (decomposed i-FST, one optimizer step per batch)',
            'value': tok_total * a.steps / el, 'unit': 'tokens/s', 'n_gpus': world, 'steps': a.steps,
            'warmup': a.warmup, 'ms_per_step': el / a.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': '{}{}: V={} S={} R={} K={}, batch {} x seqlen {} per GPU'.format(
                           desc.replace('farnn 0', 'farnn {}'.format(a.farnn)), ' with the CRF loss' if a.crf else '', V, S, R, K, B, L),
                       'valid_tokens_per_step': tok_total, 'padded_tokens_per_step': world * B * L,
                       'parallelism': 'data parallel x{}{}'.format(world, ', one RCCL all-reduce of the gradients' if world > 1 else ''),
                       'final_loss': float(loss.detach())},
            'roofline': {'bound': 'valu_f32', 'kernel': 'train_backward_kernel (+train_forward_kernel, train_loss_kernel, atb_*)',
                         'achieved': achieved, 'peak': F32_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': achieved / F32_PEAK_TFLOPS,
                         'traffic': None, 'algorithmic_flops_per_launch': flops,
                         'kernel_avg_us': lib_s * 1e6, 'launches_timed': lib_n,
                         'step_latency_us': lib_s * 1e6 / (2 * L),
                         'note': 'the library part of the step is bound by the latency of 64 sequential recurrence '
                                 'steps per direction (forward, then backward in time), not by bandwidth or flops'},
        }
        if world == 1 and not a.no_cpu_baseline and not a.crf and not a.farnn:
            out['cpu_baseline'] = train_cpu_baseline(p, beta, x, lengths, labels, a.cpu_seconds)
        print(result_line(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def train_cpu_baseline(p, beta, x, lengths, labels, budget_s):
    """The torch-fp32 training oracle (oracle/farnn_train_oracle.py, batch-vectorised form) on the host cores: whole
    steps on the same batch until the time budget is used."""
    from oracle import farnn_train_oracle as to
    q = {'S1': p['S1'], 'S2': p['S2'], 'V_embed': p['V_embed'], 'embed_r_generalized': p['G'],
         'embedding.weight': p['E'], 'C_output_mat': p['C'], 'wildcard_mat': p['W'], 'h0': p['h0'], 'hT': p['hT'],
         'beta_vec': beta}
    q = {k: v.detach().cpu() for k, v in q.items()}
    q['priority_mat'] = torch.eye(q['C_output_mat'].shape[0])
    xs, ls, lb = torch.from_numpy(x), torch.from_numpy(lengths), torch.from_numpy(labels)
    to.train_step_batched(q, xs, ls, lb, nl='tanh', additional_nonlinear='tanh')          # warm-up (thread pool, allocator)
    n, t0 = 0, time.perf_counter()
    while True:
        to.train_step_batched(q, xs, ls, lb, nl='tanh', additional_nonlinear='tanh')
        n += 1
        el = time.perf_counter() - t0
        if el + el / n > budget_s:
            break
    return {'value': int(lengths.sum()) * n / el, 'unit': 'tokens/s', 'cores': torch.get_num_threads(), 'kind': 'port',
            'sample': 'torch-fp32 autograd oracle (batch-vectorised restatement of the reference step), {} whole steps of '
                      'the same batch in {:.1f} s; the reference itself takes 0.73 s per step on 8 cores (DESIGN.md)'.format(n, el)}


# ------------------------------------------------------------------------------------------ the tagging measurement
def run_tagging(a, name, steps, warmup, world, rank, dev, dist, want_pipelined, event_stride, headline):
    """Build `name`, time `steps` steps (contract: barrier + sync on both sides, max over ranks), then -- after
    the timed region -- re-check the batch against the oracle.  Returns the result dict on rank 0, else None."""
    from re2nn_seq_amd import _lib
    B, L = a.batch, a.seqlen
    n_main = max(a.streams, 1) if headline else 1
    n_pipe = max(n_main, 2 if want_pipelined else 1)
    built = [build_workload(name, B, L, rank, a.rank, a.full_length, a.farnn, a.semiring, a.crf and name == 'decomp') for _ in range(n_pipe)]
    h, x, lengths, extras = built[0]
    handles = [b[0] for b in built]
    del built
    # the timed region rotates through NBATCH different batches (same lengths, fresh tokens: batch_variants)
    xs = batch_variants(name, x, lengths, rank, a.batches) if a.batches > 1 else [x]
    xds = [torch.from_numpy(v).to(dev) for v in xs]
    xd = xds[0]
    ld = torch.from_numpy(lengths).to(dev)
    tags_bufs = [torch.full((B, L), -7, dtype=torch.int32, device=dev) for _ in handles]
    # N > 1: the tag ids of step i are gathered (RCCL all-gather over xGMI, on RCCL's own stream) while
    # step i+1 computes (re2nn_seq_amd.dist.OverlappedGather: two blocks in rotation)
    og = None
    if world > 1:
        from re2nn_seq_amd.dist import OverlappedGather
        og = OverlappedGather(B, L, dev, comm=NATIVE_COMM[0])
    for hh in handles:
        hh.reserve(B, L)
    streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in handles[1:]]

    timed_by = ['HIP events on the dispatch packets of every N-th step (hipExtLaunchKernelGGL)']
    last_written = {}                            # output buffer -> the batch its latest step tagged

    def timed_region(n_streams, steps, warmup, stride):
        counter = [0]

        def step():
            k = counter[0] % n_streams
            xb = xds[counter[0] % len(xds)]
            last_written[k] = counter[0] % len(xds)
            counter[0] += 1
            st = streams[k]
            if world == 1:
                handles[k].tag(xb.data_ptr(), ld.data_ptr(), B, L, _lib.MODE_LOCAL, tags_bufs[k].data_ptr(),
                               None, None, st.cuda_stream)
                return
            with torch.cuda.stream(st):
                out = og.next_output()
                handles[k].tag(xb.data_ptr(), ld.data_ptr(), B, L, _lib.MODE_LOCAL, out.data_ptr(),
                               None, None, st.cuda_stream)
                og.submit()

        def drain():
            if og is not None:
                og.drain()

        for _ in range(warmup):
            step()
        drain()
        torch.cuda.synchronize(dev)
        counter[0] = 0                           # the timed steps take the batches in order: step i runs batch i mod NBATCH
        graph = None
        if a.graph > 0 and world == 1 and n_streams == 1 and stride == 0:
            # the whole step (every kernel farnn_tag enqueues) recorded once, replayed steps/N times
            side = torch.cuda.Stream(dev)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                for gi in range(a.graph):
                    handles[0].tag(xds[gi % len(xds)].data_ptr(), ld.data_ptr(), B, L, _lib.MODE_LOCAL, tags_bufs[0].data_ptr(),
                                   None, None, torch.cuda.current_stream(dev).cuda_stream)
            graph.replay()
            torch.cuda.synchronize(dev)
        # One kernel per step (the fused chain + decode launch), auto stride: ONE pair of HIP events on the launch stream
        # around the K launches of the timed region -- every launch is timed, none pays for an event pair of its own (a pair
        # costs ~6 us of stream time; with the driver's --steps 20 that was 3.6 us on every step).  Average = region / K,
        # launch gaps included (conservative).  Steps of several kernels keep per-launch events (every N-th step).
        region = (stride != 0 and a.event_stride < 0 and world == 1 and n_streams == 1 and graph is None
                  and 'fused' in handles[0].kernel_name(_lib.KERN_CHAIN))
        # A step of SEVERAL kernels in a short run (the driver's --steps 20): event pairs on every second step's dispatch packets would
        # add ~6 us of stream time per timed launch to a 31 us step.  The timed region then carries no per-launch event at all and the
        # per-kernel durations come from 8 more steps of the same loop (same handles, same rotating batches) run right behind it,
        # every launch of them timed -- outside the K timed steps, like the warm-up.
        post = (not region and stride != 0 and a.event_stride < 0 and world == 1 and n_streams == 1 and graph is None and steps < 64)
        ev = None
        for hh in handles[:n_streams]:
            hh.set_profiling(0 if (region or post) else stride)   # HIP events around the kernels of every N-th step
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        if region:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        # (nothing between the warm-up's synchronize and the region: round 6 tried collecting the interpreter's garbage here and holding the
        #  collector off inside the region -- the tens of milliseconds the GPU then sat idle made the driver's 20 steps 2-4 us SLOWER each,
        #  32.9-34.5 against 30.4-30.7 us in alternating runs on one box; running the side configurations first did not help either)
        t0 = time.perf_counter()
        if region:
            ev[0].record(streams[0])
        if graph is not None:
            assert steps % a.graph == 0, '--steps must be a multiple of --graph'
            for _ in range(steps // a.graph):
                graph.replay()
            last_written[0] = (a.graph - 1) % len(xds)
        else:
            for _ in range(steps):
                step()
        if region:
            ev[1].record(streams[0])
        drain()                                 # every gather of the K steps is inside the timed region
        torch.cuda.synchronize(dev)
        if world > 1:                           # (one rank: the synchronize above already closes the region)
            dist.barrier()
            torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        if post:
            handles[0].set_profiling(1)
            for _ in range(8):
                step()
            torch.cuda.synchronize(dev)
            timed_by[0] = ('HIP events on the dispatch packets of 8 steps run right behind the timed region (a multi-kernel step in a short '
                           'run: no event inside the {} timed steps)'.format(steps))
        sums = [0.0, 0, 0.0, 0]
        if region:
            sums[0], sums[1] = float(ev[0].elapsed_time(ev[1])), steps
            timed_by[0] = 'one HIP event pair on the launch stream around the {} launches of the timed region'.format(steps)
        for hh in handles[:n_streams]:
            ms, n = hh.kernel_time(_lib.KERN_CHAIN); sums[0] += ms; sums[1] += n
            ms, n = hh.kernel_time(_lib.KERN_SCORE); sums[2] += ms; sums[3] += n
            hh.set_profiling(0)
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, sums

    elapsed, (chain_ms, chain_n, score_ms, score_n) = timed_region(n_main, steps, warmup, event_stride)
    timed_how = timed_by[0]
    pipelined = None
    if want_pipelined and n_main == 1:
        # same K steps with two batches in flight (two streams, two handles): the score/decode kernel
        # of one batch overlaps the recurrence of the next.  Reported beside `value`, never as it.
        pel, _ = timed_region(2, steps, warmup, 0)
        pipelined = (2, pel)

    tok_local = int(lengths.sum())
    tok_min = tok_max = tok_local
    if world > 1:
        n = torch.tensor([tok_local], dtype=torch.int64, device=dev)
        dist.all_reduce(n, op=dist.ReduceOp.SUM)
        tok_total = int(n.item())
        n.fill_(tok_local); dist.all_reduce(n, op=dist.ReduceOp.MIN); tok_min = int(n.item())
        n.fill_(tok_local); dist.all_reduce(n, op=dist.ReduceOp.MAX); tok_max = int(n.item())
    else:
        tok_total = tok_local
    if rank != 0:
        for hh in handles:
            hh.close()
        return None

    desc, V, S, C = WORKLOADS[name]
    chain_avg_s = (chain_ms / max(chain_n, 1)) * 1e-3
    score_avg_s = (score_ms / max(score_n, 1)) * 1e-3
    out = {
        'value': tok_total * steps / elapsed,
        'unit': 'tokens/s',
        'steps': steps, 'warmup': warmup,
        'ms_per_step': elapsed / steps * 1e3,
        'config': {'workload': '{}: V={} S={} C={}{}, batch {} x seqlen {} per GPU, lengths {}'
                               .format(desc, V, S, C,
                                       ' R={} farnn={}{}'.format(a.rank, a.farnn, ' use_crf=1' if (a.crf and name == 'decomp') else '') if name.startswith('decomp') else '',
                                       B, L, 'all {}'.format(L) if a.full_length else 'U[5,{}]'.format(L)),
                   'valid_tokens_per_step': tok_total, 'padded_tokens_per_step': world * B * L,
                   'valid_tokens_per_rank_min_max': [tok_min, tok_max],
                   'batches_rotated': len(xs),
                   'parallelism': 'batch-sharded x{} (weights replicated{}){}'.format(
                       world, ', RCCL all_gather of tag ids' if world > 1 else '',
                       ', {} batches in flight per GPU'.format(n_main) if n_main > 1 else '')},
        'roofline': roofline_for(name, a, h, extras, tok_local, chain_avg_s, score_avg_s, chain_n),
    }
    out['roofline']['timed_by'] = timed_how
    if pipelined:
        out['pipelined'] = {'streams': pipelined[0], 'value': tok_total * steps / pipelined[1],
                            'unit': 'tokens/s', 'ms_per_step': pipelined[1] / steps * 1e3,
                            'note': 'same K steps with two batches in flight on two HIP streams'}
    if world == 1 and not a.no_parity:
        # the last timed step wrote tags_bufs[(steps - 1) % n_main] from batch (steps - 1) mod NBATCH (a graph: the last of its steps)
        last = last_written[(steps - 1) % n_main]
        got = tags_bufs[(steps - 1) % n_main].cpu().numpy()
        out['parity'] = parity_check(name, h, extras, xs[last], lengths, got, dev)
        out['parity']['batch_checked'] = last
        if out['parity']['tags_equal'] is False:
            print('WARNING: {}: GPU results differ from the oracle'.format(name), file=sys.stderr)
    if headline and world == 1 and name in ('ifst', 'synth512') and h.has_compact():
        # the compact form (bit-packed blocks + active-state walk) reported SEPARATELY; the contract
        # number above stays fp32-dense.  Same batch, same handle; its tags must equal the dense kernel's.
        dense_tags = tags_bufs[(steps - 1) % n_main].clone()
        xd = xds[last_written[(steps - 1) % n_main]]     # the batch those tags belong to
        h.set_compact(True)
        csteps = max(steps, 50) if name == 'ifst' else steps
        for _ in range(5):
            h.tag(xd.data_ptr(), ld.data_ptr(), B, L, _lib.MODE_LOCAL, tags_bufs[0].data_ptr(), None, None, streams[0].cuda_stream)
        h.set_profiling(max(1, csteps // 8))
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(csteps):
            h.tag(xd.data_ptr(), ld.data_ptr(), B, L, _lib.MODE_LOCAL, tags_bufs[0].data_ptr(), None, None, streams[0].cuda_stream)
        torch.cuda.synchronize(dev)
        cel = time.perf_counter() - t0
        cms, cn = h.kernel_time(_lib.KERN_CHAIN)
        sms, sn = h.kernel_time(_lib.KERN_SCORE)
        h.set_profiling(0)
        out['compact'] = {'value': tok_local * csteps / cel, 'unit': 'tokens/s', 'ms_per_step': cel / csteps * 1e3, 'steps': csteps,
                          'bytes_per_token': h.kernel_algorithmic_bytes(_lib.KERN_CHAIN, tok_local) / tok_local,
                          'dense_bytes_per_token': 2.0 * S * S * 4 + 8,
                          'kernel': h.kernel_name(_lib.KERN_CHAIN), 'kernel_avg_us': cms / max(cn, 1) * 1e3,
                          'score_decode_avg_us': sms / max(sn, 1) * 1e3,
                          'tags_equal_dense': bool(torch.equal(tags_bufs[0], dense_tags)),
                          'note': 'bit-packed transition blocks, walk over the non-zero state entries only; latency-bound '
                                  '(serial step chain), reported beside the fp32-dense contract number, never as it'}
        h.set_compact(False)
    if headline and world == 1 and not a.no_cpu_baseline and name == 'ifst':
        out['cpu_baseline'] = cpu_baseline(extras, x, lengths, a.cpu_seconds)
        out['cpu_baseline_faithful'] = cpu_baseline_faithful(extras, x, lengths, min(a.cpu_seconds, 6.0))
    if headline and world == 1 and name == 'ifst' and not a.no_parity:
        for hh in handles:
            hh.close()
        handles = []
        out['host_inclusive'] = host_inclusive(extras, x, lengths)
    for hh in handles:
        hh.close()
    del extras
    torch.cuda.empty_cache()
    return out


def main():
    a = parse()
    if 'WORLD_SIZE' not in os.environ and a.gpus > 1:
        return launch_ranks(a)                 # this process never initialises the GPU
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != a.gpus:
        raise SystemExit('bench.py: --gpus {} but WORLD_SIZE={} (launch with torchrun --nproc-per-node {} or '
                         'let bench.py start the ranks itself)'.format(a.gpus, world, a.gpus))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU fallback for the product path)')
    # test hook (single-GPU boxes): FARNN_BENCH_ONE_DEVICE=1 runs every rank on cuda:0 over gloo, to exercise
    # the N>1 code path where RCCL (one rank per device) cannot be used
    one_dev = os.environ.get('FARNN_BENCH_ONE_DEVICE') == '1'
    if one_dev:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if one_dev:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=dev)   # "nccl" is RCCL on ROCm
        # the N > 1 path must not limp on with a wrong topology: every rank checks what it joined
        if dist.get_world_size() != world or dist.get_rank() != rank:
            raise SystemExit('bench.py: process group has world {} / rank {}, the launcher said {} / {}'.format(
                dist.get_world_size(), dist.get_rank(), world, rank))
        if not one_dev and torch.cuda.device_count() < world:
            raise SystemExit('bench.py: {} ranks but only {} GPUs visible to rank {}'.format(world, torch.cuda.device_count(), rank))
        if rank == 0:
            try:
                ver = '.'.join(str(v) for v in torch.cuda.nccl.version())
            except Exception as e:                 # noqa: BLE001 -- informational only
                ver = 'unknown ({})'.format(type(e).__name__)
            print('bench.py: {} ranks, backend {}, RCCL {}, one device per rank: {}'.format(
                world, dist.get_backend(), ver, not one_dev), file=sys.stderr, flush=True)
    from re2nn_seq_amd import _lib  # noqa: F401
    gather_info = None
    if world > 1:
        gather_info = {'backend': 'torch.distributed all_gather_into_tensor ({})'.format(dist.get_backend())}
        if a.gather == 'native':
            # the C-ABI collective (include/farnn_rccl.h): rank 0's communicator id travels through the process group that is
            # already up; from then on no torch.distributed call is on the tagging path (OverlappedGather(comm=...)).  On the
            # one-device dry run RCCL cannot host several ranks on one GPU: the loopback communicator (gloo behind the same
            # interface) exercises everything around the collective.
            if one_dev:
                from re2nn_seq_amd.dist import LoopbackCommunicator
                NATIVE_COMM[0] = LoopbackCommunicator(world, rank)
                gather_info = {'backend': 'loopback communicator over gloo (dry run: the native path around the collective)'}
            else:
                from re2nn_seq_amd import _rccl
                box = [_rccl.unique_id() if rank == 0 else None]
                dist.broadcast_object_list(box, src=0)
                NATIVE_COMM[0] = _rccl.Communicator(box[0], world, rank, local)
                gather_info = {'backend': 'farnn_rccl_gather_tags (ncclAllGather, libfarnn_rccl.so, side HIP stream)',
                               'rccl_version_code': _rccl.lib().farnn_rccl_version()}
            n_comm = NATIVE_COMM[0].count()
            if n_comm != world:
                raise SystemExit('bench.py: the communicator has {} ranks, the launcher said {}'.format(n_comm, world))
            gather_info['comm_count'] = n_comm
        else:
            gather_info['comm_count'] = dist.get_world_size()

    if a.workload == 'train':
        return run_train(a, world, rank, dev, dist)
    stride = a.event_stride if a.event_stride >= 0 else auto_event_stride(a.steps, a.workload)
    want_pipe = not a.no_pipelined and a.streams == 1 and a.workload != 'synth512'   # no second 63 GB replica
    res = run_tagging(a, a.workload, a.steps, a.warmup, world, rank, dev, dist, want_pipe, stride, True)
    if rank == 0:
        out = {'metric': 'tagged tokens/sec @ batch=256, seqlen=64; achieved HBM GB/s vs peak',
               'value': res.pop('value'), 'unit': res.pop('unit'), 'n_gpus': world,
               'steps': res.pop('steps'), 'warmup': res.pop('warmup'), 'ms_per_step': res.pop('ms_per_step'),
               'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic'}
        out.update(res)
        if gather_info is not None:
            out['gather'] = gather_info
        defaults = (a.batch, a.seqlen, a.rank, a.farnn, a.semiring, a.full_length, a.streams, a.graph) == \
                   (256, 64, 50, 0, 'sum', False, 1, 0)
        if world == 1 and a.workload == 'ifst' and defaults and not a.no_other_configs:
            others = []
            for name, st, wu, over, label in OTHER_CONFIGS:
                t0 = time.perf_counter()
                try:
                    a2 = argparse.Namespace(**vars(a))
                    env = over.get('env', {})
                    for kk, vv in over.items():
                        if kk != 'env':
                            setattr(a2, kk, vv)
                    saved = {kk: os.environ.get(kk) for kk in env}
                    os.environ.update(env)
                    saved_wl = WORKLOADS[name]
                    if over.get('states'):
                        WORKLOADS[name] = (saved_wl[0] + ' [S = {}]'.format(over['states']), saved_wl[1], over['states'], saved_wl[3])
                    try:
                        r = run_tagging(a2, name, st, wu, 1, 0, dev, None, False, auto_event_stride(st, name), False)
                    finally:
                        WORKLOADS[name] = saved_wl
                        for kk, vv in saved.items():
                            if vv is None:
                                os.environ.pop(kk, None)
                            else:
                                os.environ[kk] = vv
                    r['workload'] = label
                    name = label
                    r['wall_s'] = time.perf_counter() - t0
                except Exception as e:          # a side measurement must not take the headline line down
                    r = {'workload': name, 'error': '{}: {}'.format(type(e).__name__, e)}
                others.append(r)
                print(other_config_line(r), flush=True)       # its own line, BEFORE the result line
            out['other_configs'] = others
        out['full'] = write_full(out)
        print(result_line(out), flush=True)                   # the ONE result line: last on stdout, <= LINE_CAP bytes
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

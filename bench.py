#!/usr/bin/env python3
"""Benchmark of the FA-RNN forward tagging path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload ifst|ifst_crf|fst4|decomp]

A "step" is one pass of the hot path (farnn_tag through the C-ABI: recurrence chain +
score/decode kernels, plus the RCCL gather of tag ids when N > 1) over one batch of synthetic
input already resident in HBM.  Default workload = BASELINE.json configs[1]: ATIS-BIO-sized
onehot i-FST (V=950, S=71, C=128), batch 256 x seqlen 64 per GPU, lengths ~ U[5,64] with one
full-length row.  Weak scaling: every rank tags its own 256-sequence
shard; `value` = valid (non-pad) tokens tagged by all ranks per second.

Rank 0 prints ONE JSON line (contract in the task statement), carrying
  roofline     the dominant kernel (chain_kernel) priced against the HBM peak: algorithmic bytes
               per launch ((2*S*S*4+12) per valid token) / its mean duration measured
               with HIP events on the launch stream inside the timed region;
  cpu_baseline the C port of the oracle (oracle/farnn_oracle.c, OpenMP over sequences) timed on
               this host's cores on the same batch (N=1, rank 0 only) -- also used to re-check
               the GPU tags after the timed region.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
F32_MFMA_PEAK_TFLOPS = 157.3   # dense f32 MFMA peak (64 FLOP/clk/SIMD; MI355X_MICROARCH.md)

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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--workload', default='ifst', choices=sorted(WORKLOADS))
    ap.add_argument('--batch', type=int, default=256, help='sequences per GPU')
    ap.add_argument('--seqlen', type=int, default=64)
    ap.add_argument('--rank', type=int, default=50, help='decomp: CP rank')
    ap.add_argument('--farnn', type=int, default=0, help='decomp: gate mode 0/1/2 (reference --farnn)')
    ap.add_argument('--semiring', default='sum', choices=['sum', 'max'], help='decomp: reference --train_mode')
    ap.add_argument('--full-length', action='store_true', help='all sequences at full length')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=10.0)
    ap.add_argument('--crf', action='store_true', help='train: CRF negative log-likelihood instead of the cross-entropy')
    ap.add_argument('--streams', type=int, default=1,
                    help='in-flight batches for the main timed region: steps alternate over this many '
                         'HIP streams, each with its own model handle and workspace')
    ap.add_argument('--no-pipelined', action='store_true', help='skip the extra 2-streams measurement')
    ap.add_argument('--graph', type=int, default=0,
                    help='N > 0: capture N consecutive steps into one HIP graph and replay it (steps must be a '
                         'multiple of N); 0 = plain stream launches')
    ap.add_argument('--event-stride', type=int, default=16,
                    help='time the kernels of every N-th step with HIP events (0 = never)')
    return ap.parse_args()


def build_workload(name, B, L, rank_id, cp_rank, full_length, farnn=0, semiring='sum'):
    """Returns (handle, x, lengths, extras).  Weights use one seed on every rank (replicated
    model); the batch is seeded per rank (each rank owns a different shard)."""
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
        del T
        torch.cuda.empty_cache()
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
        extras.update(Tf=T + W, O=O, h0=h0, hT=hT)
    elif name == 'fst4':
        T, W, O, h0, hT = synth.random_ifst_tensors(V, S, C, wrng)
        # the 4-D layout of the same automaton: label c lives on the edges into states with O[c,j]=1
        T4 = np.einsum('vsj,cj->vcsj', T, O).astype(np.float32)
        W4 = np.einsum('sj,cj->csj', W, O).astype(np.float32)
        h = _lib.create_onehot_fst4(T4, W4, h0, hT, device=dev)
        del T4
    elif name in ('decomp1', 'decomp0'):
        p = synth.random_decomposed_params(V, S, C, cp_rank, 100, wrng)
        RO = 70
        f = lambda *shape, sc=0.2: (wrng.randn(*shape) * sc).astype(np.float32)      # noqa: E731
        if name == 'decomp1':
            # the scoring GEMM of one token: U = abw[S,S] . S2o[S,RO], the a b~ scaling and br = sum S1o . U
            extras['mfma_flops_per_token'] = 2.0 * S * S * RO + 2.0 * S * S + 2.0 * S * RO
            h = _lib.create_decomp_ind1(p['V_embed'], p['S1'], p['S2'], p['wildcard_mat'], f(C, RO, sc=0.5),
                                        f(S, RO), f(S, RO), p['start_vector'], p['final_vector'], nl='tanh',
                                        semiring=semiring, device=dev)
        else:
            h = _lib.create_decomp_fst(p['V_embed'], f(C, cp_rank, sc=0.5), p['S1'], p['S2'], f(C, RO, sc=0.5),
                                       f(S, RO), f(S, RO), p['wildcard_mat'], p['start_vector'],
                                       p['final_vector'], nl='tanh', semiring=semiring, device=dev)
    else:
        p = synth.random_decomposed_params(V, S, C, cp_rank, 100, wrng)
        Vgen = p['V_embed']          # beta = 1: the generalized table is V_embed itself
        gates = None
        if farnn:
            gates = {'Wss1': wrng.randn(S, S) * 0.1, 'Wrs1': wrng.randn(cp_rank, S) * 0.1, 'bs1': np.full(S, 1.0)}
            if farnn == 2:
                gates.update(Wss2=wrng.randn(S, S) * 0.1, Wrs2=wrng.randn(cp_rank, S) * 0.1, bs2=np.full(S, 1.0))
        h = _lib.create_decomp_ifst(Vgen, p['S1'], p['S2'], p['wildcard_mat'], p['C_output_mat'],
                                    p['start_vector'], p['final_vector'], nl='tanh', farnn=farnn, gates=gates,
                                    sigmoid_exponent=5, semiring=semiring, device=dev)
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


def cpu_baseline(extras, x, lengths, gpu_tags, seconds):
    """Time the C port of the oracle on this host's cores on the same batch (bounded sample) and
    re-check the GPU tags against it.  OpenMP over the 256 sequences does not scale to every
    host's full thread count (a 256-thread box runs it slower than 32 threads), so a short sweep
    picks the fastest thread count first and `cores` reports the threads actually used."""
    from oracle import c_port
    c_port.load(native=True)
    ncpu = os.cpu_count() or 1
    args = (extras['Tf'], extras['O'], extras['h0'], extras['hT'], x, lengths)
    tags, _, _ = c_port.onehot_ifst_tag(*args, nthreads=min(ncpu, 8))      # warm-up + check
    parity = bool(np.array_equal(tags, gpu_tags))
    tok = int(lengths.sum())

    def rate(nthreads, budget):
        c_port.onehot_ifst_tag(*args, nthreads=nthreads)
        n, t0 = 0, time.perf_counter()
        while True:
            c_port.onehot_ifst_tag(*args, nthreads=nthreads)
            n += 1
            el = time.perf_counter() - t0
            if el >= budget or n >= 5000:
                return tok * n / el, n, el

    cands = sorted({c for c in (ncpu, ncpu // 2, ncpu // 4, 64, 32, 16, 8) if 1 <= c <= ncpu})
    best = max(cands, key=lambda c: rate(c, 0.6)[0])
    value, n, el = rate(best, seconds)
    return {'value': value, 'unit': 'tokens/s', 'cores': int(best), 'kind': 'port',
            'sample': '{} passes of the same {}x{} batch ({} valid tokens) in {:.1f} s; C port of the '
                      'oracle (T+W hoisted, OpenMP over sequences), best of thread counts {} on a '
                      '{}-thread host'.format(n, x.shape[0], x.shape[1], tok, el, cands, ncpu)}, parity


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
        # algorithmic bytes of the library part per valid token: the state stash, the per-token adjoint rows and
        # the score-side rows, each written once and read once
        alg = (2.0 * (16 * S + 6 * R + 2 * K) * 4) * tok_local
        lib_s = (lib_ms / max(lib_n, 1)) * 1e-3
        achieved = alg / lib_s / 1e9 if lib_s > 0 else 0.0
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
            'roofline': {'bound': 'hbm', 'kernel': 'train_backward_kernel (+train_forward_kernel, train_loss_kernel, atb_*)',
                         'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                         'traffic': None, 'algorithmic_bytes_per_launch': alg,
                         'kernel_avg_us': lib_s * 1e6, 'launches_timed': lib_n,
                         'note': 'the library part of the step is bound by the latency of 64 sequential recurrence '
                                 'steps per direction, not by bandwidth'},
        }
        if world == 1 and not a.no_cpu_baseline and not a.crf and not a.farnn:
            out['cpu_baseline'] = train_cpu_baseline(p, beta, x, lengths, labels, a.cpu_seconds)
        print(json.dumps(out), flush=True)
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


def main():
    a = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
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
    from re2nn_seq_amd import _lib

    B, L = a.batch, a.seqlen
    if a.workload == 'train':
        return run_train(a, world, rank, dev, dist)
    if a.workload == 'synth512':
        a.no_pipelined = True            # a second 63 GB replica of the weights is pointless here
    h, x, lengths, extras = build_workload(a.workload, B, L, rank, a.rank, a.full_length, a.farnn, a.semiring)
    n_pipe = max(a.streams, 1 if a.no_pipelined else 2)
    handles = [h] + [build_workload(a.workload, B, L, rank, a.rank, a.full_length, a.farnn, a.semiring)[0]
                     for _ in range(n_pipe - 1)]
    xd = torch.from_numpy(x).to(dev)
    ld = torch.from_numpy(lengths).to(dev)
    tags_bufs = [torch.empty((B, L), dtype=torch.int32, device=dev) for _ in handles]
    tags = tags_bufs[0]
    # N > 1: the tag ids of step i are gathered (RCCL all-gather over xGMI, on RCCL's own stream) while
    # step i+1 computes (re2nn_seq_amd.dist.OverlappedGather: two blocks in rotation)
    og = None
    if world > 1:
        from re2nn_seq_amd.dist import OverlappedGather
        og = OverlappedGather(B, L, dev)
    for hh in handles:
        hh.reserve(B, L)
    streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in handles[1:]]

    def timed_region(n_streams, steps, warmup, event_stride):
        """barrier + sync, `steps` steps alternating over `n_streams` streams, sync + barrier;
        returns (elapsed max-over-ranks is taken by the caller, per-kernel event sums)."""
        counter = [0]

        def step():
            k = counter[0] % n_streams
            counter[0] += 1
            st = streams[k]
            if world == 1:
                handles[k].tag(xd.data_ptr(), ld.data_ptr(), B, L, _lib.MODE_LOCAL, tags_bufs[k].data_ptr(),
                               None, None, st.cuda_stream)
                return
            with torch.cuda.stream(st):
                out = og.next_output()
                handles[k].tag(xd.data_ptr(), ld.data_ptr(), B, L, _lib.MODE_LOCAL, out.data_ptr(),
                               None, None, st.cuda_stream)
                og.submit()

        def drain():
            if og is not None:
                og.drain()

        for _ in range(warmup):
            step()
        drain()
        torch.cuda.synchronize(dev)
        graph = None
        if a.graph > 0 and world == 1 and n_streams == 1 and event_stride == 0:
            # the whole step (every kernel farnn_tag enqueues) recorded once, replayed steps/N times
            side = torch.cuda.Stream(dev)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                for _ in range(a.graph):
                    handles[0].tag(xd.data_ptr(), ld.data_ptr(), B, L, _lib.MODE_LOCAL, tags_bufs[0].data_ptr(),
                                   None, None, torch.cuda.current_stream(dev).cuda_stream)
            graph.replay()
            torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        for hh in handles[:n_streams]:
            hh.set_profiling(event_stride)     # HIP events around the kernels of every N-th step
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        if graph is not None:
            assert steps % a.graph == 0, '--steps must be a multiple of --graph'
            for _ in range(steps // a.graph):
                graph.replay()
        else:
            for _ in range(steps):
                step()
        drain()                                 # every gather of the K steps is inside the timed region
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        sums = [0.0, 0, 0.0, 0]
        for hh in handles[:n_streams]:
            ms, n = hh.kernel_time(_lib.KERN_CHAIN); sums[0] += ms; sums[1] += n
            ms, n = hh.kernel_time(_lib.KERN_SCORE); sums[2] += ms; sums[3] += n
            hh.set_profiling(0)
        if world > 1:
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        return el, sums

    elapsed, (chain_ms, chain_n, score_ms, score_n) = timed_region(a.streams, a.steps, a.warmup, a.event_stride)
    pipelined = None
    if not a.no_pipelined and a.streams == 1:
        # same K steps with two batches in flight (two streams, two handles): the score/decode kernel
        # of one batch overlaps the recurrence of the next.  Reported beside `value`, never as it.
        pel, _ = timed_region(2, a.steps, a.warmup, 0)
        pipelined = (2, pel)

    tok_local = int(lengths.sum())
    if world > 1:
        n = torch.tensor([tok_local], dtype=torch.int64, device=dev)
        dist.all_reduce(n, op=dist.ReduceOp.SUM)
        tok_total = int(n.item())
    else:
        tok_total = tok_local

    if rank == 0:
        desc, V, S, C = WORKLOADS[a.workload]
        # the dominant kernel of this workload (by time) is the one priced against the roofline
        chain_avg_s = (chain_ms / max(chain_n, 1)) * 1e-3
        score_avg_s = (score_ms / max(score_n, 1)) * 1e-3
        dom = _lib.KERN_SCORE if score_avg_s > chain_avg_s else _lib.KERN_CHAIN
        dom_avg_s = score_avg_s if dom == _lib.KERN_SCORE else chain_avg_s
        alg_bytes = h.kernel_algorithmic_bytes(dom, tok_local)
        achieved = alg_bytes / dom_avg_s / 1e9 if dom_avg_s > 0 else 0.0
        bound, peak, unit = 'hbm', HBM_PEAK_GBS, 'GB/s'
        if dom == _lib.KERN_SCORE and 'mfma_flops_per_token' in extras and 'mfma' in h.kernel_name(dom):
            # the per-token scoring GEMM on the f32 matrix cores: priced against the dense f32 MFMA peak
            bound, peak, unit = 'mfma', F32_MFMA_PEAK_TFLOPS, 'TFLOP/s'
            alg_bytes = extras['mfma_flops_per_token'] * tok_local          # algorithmic FLOPs per launch
            achieved = alg_bytes / dom_avg_s / 1e12 if dom_avg_s > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
        if os.path.exists(tpath):
            with open(tpath) as f:
                traffic = json.load(f).get(a.workload, {}).get('hbm_bytes_per_launch')
        out = {
            'metric': 'tagged tokens/sec @ batch=256, seqlen=64; achieved HBM GB/s vs peak',
            'value': tok_total * a.steps / elapsed,
            'unit': 'tokens/s',
            'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': elapsed / a.steps * 1e3,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': '{}: V={} S={} C={}, batch {} x seqlen {} per GPU, lengths {}'
                                   .format(desc, V, S, C, B, L,
                                           'all {}'.format(L) if a.full_length else 'U[5,{}]'.format(L)),
                       'valid_tokens_per_step': tok_total, 'padded_tokens_per_step': world * B * L,
                       'parallelism': 'batch-sharded x{} (weights replicated{}){}'.format(
                           world, ', RCCL all_gather of tag ids' if world > 1 else '',
                           ', {} batches in flight per GPU'.format(a.streams) if a.streams > 1 else '')},
            'roofline': {'bound': bound, 'kernel': h.kernel_name(dom),
                         'achieved': achieved, 'peak': peak, 'unit': unit,
                         'frac': achieved / peak, 'traffic': traffic,
                         'algorithmic_bytes_per_launch' if bound == 'hbm' else 'algorithmic_flops_per_launch': alg_bytes,
                         'kernel_avg_us': dom_avg_s * 1e6, 'launches_timed': chain_n,
                         'chain_avg_us': chain_avg_s * 1e6, 'score_decode_avg_us': score_avg_s * 1e6},
        }
        if pipelined:
            out['pipelined'] = {'streams': pipelined[0], 'value': tok_total * a.steps / pipelined[1],
                                'unit': 'tokens/s', 'ms_per_step': pipelined[1] / a.steps * 1e3,
                                'note': 'same K steps with two batches in flight on two HIP streams'}
        if world == 1 and not a.no_cpu_baseline and 'Tf' in extras and a.workload == 'ifst':
            cb, parity = cpu_baseline(extras, x, lengths, tags.cpu().numpy(), a.cpu_seconds)
            out['cpu_baseline'] = cb
            out['parity_vs_cpu_port'] = parity
            if not parity:
                print('WARNING: GPU tags differ from the CPU port of the oracle', file=sys.stderr)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

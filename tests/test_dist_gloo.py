"""The N>1 path on CPU: world_size-2 gloo.  Every rank tags its contiguous slice of the batch and
one all-gather returns the tag ids (re2nn_seq_amd/dist.py); the per-shard tagger here is the CPU
oracle (this is a test), the collective and the sharding arithmetic are the product code."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, B, out_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, 'tests'))
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from oracle import farnn_oracle as fo
    from re2nn_seq_amd import dist as fdist, synth
    rng = np.random.RandomState(11)                      # same weights and batch on every rank
    V, S, C, L = 40, 9, 6, 10
    T, W, O, h0, hT = synth.random_ifst_tensors(V, S, C, rng, edges_per_word=5.0)
    x, lengths = synth.random_batch(V, B, L, rng, min_len=1)

    def tag_fn(xs, ls):
        if xs.shape[0] == 0:
            return torch.zeros((0, L), dtype=torch.int32)
        sc = fo.onehot_ifst_scores(T, W, O, h0, hT, xs.numpy(), ls.numpy())
        return torch.from_numpy(fo.decode_argmax(sc, 0.5, 0).astype(np.int32))

    xt, lt = torch.from_numpy(x), torch.from_numpy(lengths)
    lo, hi = fdist.shard_bounds(B, rank, world)
    assert fdist.shard_batch(xt, lt)[2] == (lo, hi)
    tags = fdist.tag_sharded(tag_fn, xt, lt)                         # length-balanced shares (the default)
    tags_c = fdist.tag_sharded(tag_fn, xt, lt, balance=False)        # contiguous slices
    full = fo.decode_argmax(fo.onehot_ifst_scores(T, W, O, h0, hT, x, lengths), 0.5, 0)
    ok = tags.shape == (B, L) and np.array_equal(tags.numpy().astype(np.int64), full)
    ok = ok and np.array_equal(tags_c.numpy().astype(np.int64), full)
    with open(os.path.join(out_dir, 'rank{}.txt'.format(rank)), 'w') as f:
        f.write('ok' if ok else 'mismatch')
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('B', [8, 7, 1])
def test_sharded_tagging_two_ranks_gloo(tmp_path, B):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), B, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert (tmp_path / 'rank{}.txt'.format(r)).read_text() == 'ok'


@pytest.mark.parametrize('B', [10, 2])
def test_sharded_tagging_three_ranks_gloo(tmp_path, B):
    """world size 3: ragged shares, and a batch smaller than the world (one rank tags nothing)"""
    world = 3
    mp.spawn(_worker, args=(world, _free_port(), B, str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert (tmp_path / 'rank{}.txt'.format(r)).read_text() == 'ok'


def test_balanced_assignment_evens_out_tokens_and_long_chains():
    """ranks should finish together -- a rank's time is its longest chain, then its token count."""
    from re2nn_seq_amd.dist import balanced_assignment, shard_bounds
    rng = np.random.RandomState(5)
    for n, w in ((256, 8), (1024, 8), (257, 3), (5, 8), (0, 2)):
        lengths = torch.from_numpy(np.sort(rng.randint(1, 65, size=n))[::-1].copy())     # sorted: the worst case for slices
        assign = balanced_assignment(lengths, w)
        assert sorted(int(i) for a in assign for i in a) == list(range(n))                # a partition of the batch
        sizes = [int(a.shape[0]) for a in assign]
        assert max(sizes) - min(sizes) <= 1
        if n >= 4 * w:
            tok = [int(lengths[a].sum()) for a in assign]
            longest = [int(lengths[a].max()) for a in assign]
            assert max(tok) - min(tok) <= 64 + 0.02 * max(tok)                            # within one sequence + 2 %
            assert max(longest) - min(longest) <= 1 + (64 * w) // n
            contiguous = [int(lengths[slice(*shard_bounds(n, r, w))].sum()) for r in range(w)]
            assert max(contiguous) - min(contiguous) > 4 * (max(tok) - min(tok))          # what the slices would have given


def test_shard_bounds_cover_the_batch():
    from re2nn_seq_amd.dist import shard_bounds
    for n in (0, 1, 7, 8, 256, 8192):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _overlap_worker(rank, world, port, out_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from re2nn_seq_amd.dist import OverlappedGather
    B, L, steps = 5, 7, 9
    og = OverlappedGather(B, L, torch.device('cpu'))
    ok = True
    seen = []
    for i in range(steps):
        out = og.next_output()
        out.fill_(100 * i + rank)                 # "tagging" step i on this rank
        og.submit()
        seen.append(og.last())                    # gather buffers rotate: keep a reference per step
        if i >= 1:                                # the previous step's gather may still be in flight...
            pass
    og.drain()                                    # ...but after drain every one is complete
    last = og.last()
    for r in range(world):
        ok = ok and bool((last[r * B:(r + 1) * B] == 100 * (steps - 1) + r).all())
    # the buffer of the step before last still holds that step's gather (depth 2)
    prev = seen[-2]
    for r in range(world):
        ok = ok and bool((prev[r * B:(r + 1) * B] == 100 * (steps - 2) + r).all())
    with open(os.path.join(out_dir, 'ov{}.txt'.format(rank)), 'w') as f:
        f.write('ok' if ok else 'mismatch')
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_gather_two_ranks_gloo(tmp_path):
    """bench.py's N>1 loop: async all-gather of step i while step i+1 writes the other block."""
    world = 2
    mp.spawn(_overlap_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert (tmp_path / 'ov{}.txt'.format(r)).read_text() == 'ok'


@pytest.mark.parametrize('world', [2, 3, 4, 5, 6, 7, 8])
def test_native_gather_path_padding_and_unpermutation_world_2_to_8(world):
    """`dist.gather_tags_balanced_native` -- the code AROUND the C-ABI collective (include/farnn_rccl.h): ragged shares padded
    to the largest, rank-major blocks, ONE inverse permutation back to batch order -- at world sizes 2..8 with the loopback
    communicator (a host all-gather behind `_rccl.Communicator`'s interface), so that the first 8-GPU run of
    `bench.py --gather native` exercises RCCL and nothing else for the first time."""
    from re2nn_seq_amd import dist as fdist
    rng = np.random.RandomState(100 + world)
    L = 9
    for n in (world * 4, world * 4 + 3, world + 1, max(1, world - 1), 1, 257):
        lengths = torch.from_numpy(rng.randint(1, L + 1, size=n).astype(np.int64))
        # "tags" that name their own batch row and position: any misplaced row shows
        full = (torch.arange(n, dtype=torch.int32)[:, None] * 100 + torch.arange(L, dtype=torch.int32)[None, :])
        assign = fdist.balanced_assignment(lengths, world)
        comms = fdist.LoopbackCommunicator.board(world)
        biggest = max(int(a.shape[0]) for a in assign)
        locals_ = [full.index_select(0, assign[r]) for r in range(world)]
        for r in range(world):                                       # what every rank hands to the collective
            blk = locals_[r]
            if blk.shape[0] < biggest:
                blk = torch.cat([blk, torch.full((biggest - blk.shape[0], L), -1, dtype=torch.int32)], 0)
            comms[r].post(blk.contiguous())
        for r in range(world):
            got = fdist.gather_tags_balanced_native(locals_[r], assign, n, comms[r])
            assert got.shape == (n, L) and torch.equal(got, full), (world, n, r)
            assert comms[r].count() == world


def _native_overlap_worker(rank, world, port, out_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from re2nn_seq_amd.dist import OverlappedGather, LoopbackCommunicator
    B, L, steps = 4, 6, 7
    comm = LoopbackCommunicator(world, rank)              # gloo behind the native communicator's interface
    og = OverlappedGather(B, L, torch.device('cpu'), comm=comm)
    ok = og.world == world and comm.count() == world
    for i in range(steps):
        out = og.next_output()
        out.fill_(1000 * i + rank)
        og.submit()
    og.drain()
    last = og.last()
    for r in range(world):
        ok = ok and bool((last[r * B:(r + 1) * B] == 1000 * (steps - 1) + r).all())
    with open(os.path.join(out_dir, 'nov{}.txt'.format(rank)), 'w') as f:
        f.write('ok' if ok else 'mismatch')
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_overlapped_gather_over_a_communicator_object(tmp_path, world):
    """`OverlappedGather(comm=...)`: bench.py's `--gather native` loop with the collective behind the communicator interface
    (here the loopback one over gloo; on the GPU box `_rccl.Communicator`)."""
    mp.spawn(_native_overlap_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert (tmp_path / 'nov{}.txt'.format(r)).read_text() == 'ok'

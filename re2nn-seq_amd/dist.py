"""Batch sharding over the GPUs of one node: the only parallel axis of the tagging path.

Sequences are independent, so each rank (one process per GPU, torch.distributed over RCCL/xGMI;
"nccl" IS RCCL on ROCm) tags its share of the batch -- length-balanced by default (balanced_assignment),
contiguous slices on request -- with its own replica of the weights, and ONE collective returns the tag ids: an all-gather of int32 [B/N, L] -- 512 KiB per
GPU at the largest BASELINE config, latency- not bandwidth-bound on the 7 point-to-point xGMI
links.  The reference has no multi-GPU facility at all; this is the MI355X-native
addition of section 8e.  Works unchanged on CPU tensors with the gloo backend (tests).
"""
import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_bounds(n, rank, world_size):
    """Contiguous, balanced slice [lo, hi) of n sequences for `rank` (the first n % world_size
    ranks get one more)."""
    base, extra = divmod(n, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def balanced_assignment(lengths, world_size):
    """Length-balanced sharding ("length-sorted or length-bucketed sharding so ranks finish together"): the
    sequences, longest first (ties by index), are dealt to the ranks in serpentine order -- 0..W-1, W-1..0, ... -- so every
    rank gets the same count (+-1), a long sequence of every length class and nearly the same number of valid tokens (a
    rank's time is its longest chain, then its token count).  Deterministic: every rank derives the same assignment from
    the replicated `lengths`.  Returns a list of W int64 index tensors (rank r tags x[idx[r]])."""
    n = int(lengths.shape[0])
    lens = lengths.detach().to('cpu', torch.int64)                 # (one device sync; everything after it is vectorised)
    order = torch.sort(lens, descending=True, stable=True).indices           # longest first, ties by index
    i = torch.arange(n, dtype=torch.int64)
    rnd, pos = torch.div(i, world_size, rounding_mode='floor'), i % world_size
    rank_of = torch.where(rnd % 2 == 0, pos, world_size - 1 - pos)           # the rank that is dealt the i-th longest sequence
    return [torch.sort(order[rank_of == r]).values for r in range(world_size)]


def shard_batch_balanced(x, lengths, rank=None, world_size=None):
    """This rank's share of a replicated batch under balanced_assignment: (x_r, lengths_r, assignment)."""
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    assign = balanced_assignment(lengths, world_size)
    idx = assign[rank].to(x.device)
    return x.index_select(0, idx), lengths.index_select(0, idx), assign


def gather_tags_balanced(local_tags, assign, n_total, group=None):
    """All-gather the per-rank blocks of a balanced assignment and undo the permutation: [n_total, L], rows in batch order."""
    w = len(assign)
    if w == 1:
        return local_tags
    L = local_tags.shape[1]
    biggest = max(int(a.shape[0]) for a in assign)
    buf = local_tags
    if buf.shape[0] < biggest:
        pad = torch.full((biggest - buf.shape[0], L), -1, dtype=buf.dtype, device=buf.device)
        buf = torch.cat([buf, pad], dim=0)
    out = torch.empty((w * biggest, L), dtype=buf.dtype, device=buf.device)
    dist.all_gather_into_tensor(out, buf.contiguous(), group=group)
    # ONE inverse-permutation gather instead of a copy per rank: row j of the batch sits at out[src[j]]
    src = torch.empty((n_total,), dtype=torch.int64)
    for r, a in enumerate(assign):
        src[a] = r * biggest + torch.arange(int(a.shape[0]), dtype=torch.int64)
    return out.index_select(0, src.to(buf.device))


def gather_tags_balanced_native(local_tags, assign, n_total, comm):
    """gather_tags_balanced over the C-ABI collective (include/farnn_rccl.h, `_rccl.Communicator`) instead of
    torch.distributed: the same padded blocks, the same inverse permutation."""
    w = len(assign)
    if w == 1:
        return local_tags
    L = local_tags.shape[1]
    biggest = max(int(a.shape[0]) for a in assign)
    buf = local_tags
    if buf.shape[0] < biggest:
        pad = torch.full((biggest - buf.shape[0], L), -1, dtype=buf.dtype, device=buf.device)
        buf = torch.cat([buf, pad], dim=0)
    out = comm.gather_tags(buf.contiguous())
    src = torch.empty((n_total,), dtype=torch.int64)
    for r, a in enumerate(assign):
        src[a] = r * biggest + torch.arange(int(a.shape[0]), dtype=torch.int64)
    return out.index_select(0, src.to(buf.device))


def shard_batch(x, lengths, rank=None, world_size=None):
    r, w = world()
    rank = r if rank is None else rank
    world_size = w if world_size is None else world_size
    lo, hi = shard_bounds(x.shape[0], rank, world_size)
    return x[lo:hi], lengths[lo:hi], (lo, hi)


def gather_tags(local_tags, n_total, group=None):
    """All-gather the per-rank tag blocks [n_r, L] into [n_total, L] (same on every rank).
    Ragged shards (n_total % world != 0) are padded to the largest shard for the collective."""
    _, w = world()
    if w == 1:
        return local_tags
    L = local_tags.shape[1]
    sizes = [shard_bounds(n_total, r, w) for r in range(w)]
    biggest = max(hi - lo for lo, hi in sizes)
    buf = local_tags
    if buf.shape[0] < biggest:
        pad = torch.full((biggest - buf.shape[0], L), -1, dtype=buf.dtype, device=buf.device)
        buf = torch.cat([buf, pad], dim=0)
    out = torch.empty((w * biggest, L), dtype=buf.dtype, device=buf.device)
    dist.all_gather_into_tensor(out, buf.contiguous(), group=group)
    if all(hi - lo == biggest for lo, hi in sizes):
        return out
    return torch.cat([out[r * biggest: r * biggest + (hi - lo)] for r, (lo, hi) in enumerate(sizes)], dim=0)


def tag_sharded(tag_fn, x, lengths, group=None, balance=True):
    """Tag a replicated batch: every rank runs `tag_fn(x_shard, lengths_shard) -> int32 [n_r, L]`
    on its share; returns the gathered [B, L] tags (batch order) on every rank.  balance=True: length-balanced shares
    (balanced_assignment); False: contiguous slices."""
    if balance:
        xs, ls, assign = shard_batch_balanced(x, lengths)
        return gather_tags_balanced(tag_fn(xs, ls), assign, x.shape[0], group=group)
    xs, ls, _ = shard_batch(x, lengths)
    return gather_tags(tag_fn(xs, ls), x.shape[0], group=group)


class LoopbackCommunicator:
    """A communicator with `_rccl.Communicator`'s interface (`nranks`, `rank`, `count()`, `gather_tags(local, out, stream)`) whose
    all-gather is a host-side exchange -- over torch.distributed (gloo on CPU tensors: the multi-process tests) when a
    process group is up, else through a board shared by the instances of ONE process (`LoopbackCommunicator.board(n)`:
    every rank's instance posts its block, the gather concatenates them -- the single-process unit tests of the native
    path's padding and un-permutation at world sizes 2..8).  Test infrastructure for the code AROUND the collective; the
    collective itself is RCCL's (`_rccl.Communicator`)."""

    def __init__(self, nranks, rank, board=None, group=None):
        self.nranks, self.rank, self._board, self._group = nranks, rank, board, group

    @staticmethod
    def board(nranks):
        posted = {}
        return [LoopbackCommunicator(nranks, r, board=posted) for r in range(nranks)]

    def count(self):
        return self.nranks

    def post(self, local):
        """single-process form: every rank posts its block before anybody gathers"""
        self._board[self.rank] = local

    def gather_tags(self, local, out=None, stream=None):
        rows, L = local.shape
        if out is None:
            out = torch.empty((self.nranks * rows, L), dtype=local.dtype, device=local.device)
        if self._board is not None:
            self._board[self.rank] = local
            assert len(self._board) == self.nranks, 'post() every rank\'s block first'
            for r in range(self.nranks):
                assert tuple(self._board[r].shape) == (rows, L), 'ranks disagree on the padded block shape'
                out[r * rows:(r + 1) * rows].copy_(self._board[r])
            return out
        dist.all_gather_into_tensor(out, local.contiguous(), group=self._group)
        return out

    def close(self):
        pass


class OverlappedGather:
    """Weak-scaling serving loop: the tag ids of step i travel (one RCCL all-gather, on a stream of its own)
    while step i+1 computes.  Two output blocks and two gather buffers rotate; a block is
    handed out again only after the gather that read it has completed.

        og = OverlappedGather(B, L, device)             # torch.distributed (backend "nccl" = RCCL), or
        og = OverlappedGather(B, L, device, comm=c)     # the C-ABI collective (include/farnn_rccl.h, `_rccl.Communicator`)
        for step in ...:
            out = og.next_output()          # int32 [B, L] to tag into (waits for its previous gather)
            tagger(out)                     # enqueue the kernels on the current stream
            og.submit()                     # async all-gather of `out`
        og.drain()                          # every gather done; og.last() = [world*B, L] of the last step

    comm: the gather is `comm.gather_tags(out, gathered, stream)` on a side HIP stream that waits for the tagging stream's
    event; completion is an event of that stream (no torch.distributed anywhere on the path).
    """

    def __init__(self, B, L, device, group=None, depth=2, comm=None):
        self.comm = comm
        if comm is not None:
            self.world = comm.nranks
        else:
            _, self.world = world()
        self.group = group
        self.device = torch.device(device)
        self.out = [torch.empty((B, L), dtype=torch.int32, device=device) for _ in range(depth)]
        self.gathered = [torch.empty((self.world * B, L), dtype=torch.int32, device=device) for _ in range(depth)]
        self.works = [None] * depth
        self.i = 0
        self.cur = 0
        self.side = None
        if comm is not None and self.device.type == 'cuda':
            self.side = torch.cuda.Stream(self.device)

    def _wait(self, k):
        w = self.works[k]
        if w is None:
            return
        if self.comm is not None:
            if w is not True:
                w.synchronize()                  # (a CUDA event of the gather's stream)
        else:
            w.wait()
        self.works[k] = None

    def next_output(self):
        self.cur = self.i % len(self.out)
        self.i += 1
        self._wait(self.cur)
        return self.out[self.cur]

    def submit(self):
        k = self.cur
        if self.world == 1 and self.comm is None:
            self.gathered[k].copy_(self.out[k])
            return
        if self.comm is not None:
            if self.side is None:                # CPU tensors (tests): the exchange is synchronous
                self.comm.gather_tags(self.out[k], self.gathered[k])
                self.works[k] = True
                return
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(self.device))
            self.side.wait_event(ready)
            with torch.cuda.stream(self.side):     # (a loopback communicator's torch ops land on the side stream too)
                self.comm.gather_tags(self.out[k], self.gathered[k], self.side.cuda_stream)
            done = torch.cuda.Event()
            done.record(self.side)
            self.works[k] = done
            return
        self.works[k] = dist.all_gather_into_tensor(self.gathered[k], self.out[k], group=self.group, async_op=True)

    def drain(self):
        for k in range(len(self.works)):
            self._wait(k)

    def last(self):
        return self.gathered[self.cur]

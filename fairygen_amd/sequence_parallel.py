"""Latent-temporal token sharding of the DiT forward across the GPUs of one node (RCCL over xGMI).

Specification = the reference's USP path (``utils/xfuser/xdit_context_parallel.py:57-146`` and
``pipelines/wan_video.py:1224-1227,1310-1315,1379-1382``): tokens are frame-major, so a contiguous token
range IS a latent-temporal shard; every rank runs all 30 blocks on its N/P tokens, everything except
self-attention is token-local, and the head output is all-gathered at the end.  Wan's attention is full 3-D, so
the exact "halo" of a temporal shard is the whole sequence; two exact exchanges are built:

* ``attn_mode="allgather"``: K and V (after RMSNorm+RoPE) are all-gathered once per layer (2 collectives, 7/8 of
  2·N·3072·2 B received per rank: 293 MB at N = 27 280, P = 8); attention = local queries x all keys, all heads.
* ``attn_mode="ulysses"`` (what xfuser does, :125-146): one all-to-all turns the token shard of q/k/v into a
  head shard (all N tokens, 24/P heads), attention runs per head group over the whole sequence, a second
  all-to-all brings the output back (2 collectives, 4·(N/P)·3072·2 B·(P-1)/P sent per rank: 73 MB at P = 8 —
  a quarter of the all-gather traffic, and on the fully connected xGMI mesh every peer pair uses its own link).

* ``attn_mode="windows"`` — NOT exact: the reference's sliding-window mode (``TemporalTiler_BCTHW``,
  ``pipelines/wan_video.py:1069-1118``; ``sliding_window_size=`` / ``sliding_window_stride=`` on the call) with its overlapping
  windows of latent frames dealt to the ranks: latent-temporal shards whose overlap frames are the halo, no exchange inside
  a forward, one all-gather of the window outputs per forward, the reference's ramp blend on every rank.  Results equal the
  single-GPU sliding-window call bit for bit, and differ from the default (full 3-D attention) call like the reference's do.

On top of either, the two CFG branches of a denoise step are independent until the combine
(``pipelines/wan_video.py:296-301``), so ``ParallelLayout(cfg_parallel=2)`` gives each half of the ranks one
branch (sequence-parallel inside the half) and exchanges the two predictions with ONE world all-gather per step.

One process per GPU; process groups are torch.distributed's ("nccl" == RCCL on ROCm; "gloo" in CPU tests and in
the several-ranks-on-one-GPU rehearsal, where device tensors are staged through the host).
"""
import os

import torch
import torch.distributed as dist

ATTN_MODES = ("allgather", "ulysses", "windows")


def force_collectives():
    """FAIRYGEN_FORCE_COLLECTIVES=1 (tests / single-GPU rehearsal only): a 1-rank group does NOT take the "nothing to
    exchange" short-cuts, so every collective of the sharded path (all_gather_into_tensor, all_to_all_single, broadcast
    on device tensors, async_op and the interleaved generators) executes on a 1-rank RCCL communicator exactly as it
    does on 8 — the only way to execute those branches on a one-GPU box."""
    return os.environ.get("FAIRYGEN_FORCE_COLLECTIVES", "0") == "1"


def _staged(t, group):
    """True when `t` lives on the device but the group's backend has no device collectives (gloo rehearsal)."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


class TokenShard:
    def __init__(self, group=None, attn_mode="allgather"):
        if attn_mode not in ATTN_MODES:
            raise ValueError(f"attn_mode must be one of {ATTN_MODES}, got {attn_mode!r}")
        self.group, self.attn_mode = group, attn_mode
        if dist.is_available() and dist.is_initialized():
            self.world_size, self.rank = dist.get_world_size(group), dist.get_rank(group)
        else:
            self.world_size, self.rank = 1, 0
        # active: the exchanges run (more than one rank, or forced on an initialised 1-rank group)
        self.active = self.world_size > 1 or (force_collectives() and dist.is_available() and dist.is_initialized())

    def chunk(self, n):
        """Rows per rank: ceil split like torch.chunk (wan_video.py:1312); trailing ranks may be short/empty."""
        return (n + self.world_size - 1) // self.world_size

    def local_range(self, n, rank=None):
        rank = self.rank if rank is None else rank
        size = self.chunk(n)
        lo = min(rank * size, n)
        return lo, min(lo + size, n)

    def _padded(self, local, size):
        """(rows<=size, C) -> contiguous (size, C); short ranks are zero padded (only a suffix is padding)."""
        if local.shape[0] == size and local.is_contiguous():
            return local
        buf = torch.zeros((size, local.shape[-1]), dtype=local.dtype, device=local.device)
        buf[: local.shape[0]].copy_(local)
        return buf

    def _gather_rows(self, local, size):
        """local (rows<=size, C) -> (world*size, C)."""
        buf = self._padded(local, size)
        full = torch.empty((self.world_size * size, local.shape[-1]), dtype=local.dtype, device=local.device)
        if _staged(local, self.group):
            host = torch.empty(full.shape, dtype=full.dtype)
            dist.all_gather_into_tensor(host, buf.cpu(), group=self.group)
            full.copy_(host)
        else:
            dist.all_gather_into_tensor(full, buf, group=self.group)
        return full

    # ------------------------------------------------------------------ K/V all-gather exchange
    def all_gather_kv(self, k, v, n=None):
        """k, v (1, n_local, C) of this rank's tokens -> (1, N, C) of all tokens, in token order."""
        if not self.active:
            return k, v
        if n is None:
            counts = torch.tensor([k.shape[1]], device=k.device)
            dist.all_reduce(counts, group=self.group)
            n = int(counts.item())
        size = self.chunk(n)
        kf = self._gather_rows(k[0], size)[:n].unsqueeze(0)
        vf = self._gather_rows(v[0], size)[:n].unsqueeze(0)
        return kf, vf

    def all_gather_kv_async(self, k, v, n):
        """Start the K/V all-gather without blocking the compute stream; `.wait()` -> (k_full, v_full).

        The collectives run on the process group's own stream, so whatever the caller enqueues between this call
        and wait() — in the pipeline: the other CFG branch's GEMMs / attention — overlaps the xGMI transfer."""
        return _PendingKV(self, k, v, n)

    def all_gather_tokens(self, x, n):
        """x (1, n_local, C) -> (1, N, C) (final head all-gather, wan_video.py:1379-1382)."""
        if not self.active:
            return x
        return self._gather_rows(x[0], self.chunk(n))[:n].unsqueeze(0).contiguous()

    # ------------------------------------------------------------------ Ulysses (head <-> token) exchange
    def heads_local(self, num_heads):
        if num_heads % self.world_size != 0:
            raise ValueError(f"attn_mode='ulysses' needs num_heads ({num_heads}) divisible by the group size "
                             f"({self.world_size}); use attn_mode='allgather'")
        return num_heads // self.world_size

    def ulysses_qkv_async(self, q, k, v, n, num_heads):
        """q, k, v (1, n_local, H*D) of this rank's tokens (v may be a column slice of a wider buffer).  Starts ONE
        all-to-all; `.wait()` -> (q, k, v) as (1, N, (H/P)*D) views (row stride 3*(H/P)*D) of ALL tokens for this
        rank's head group [rank*H/P, (rank+1)*H/P)."""
        p, size = self.world_size, self.chunk(n)
        g = q.shape[-1] // num_heads * self.heads_local(num_heads)
        n_loc = q.shape[1]
        send = self.ulysses_send_buffer(n, q.shape[-1], q, n_loc)
        for j, t in enumerate((q, k, v)):
            send[:, :n_loc, j].copy_(t[0].unflatten(-1, (p, g)).transpose(0, 1))
        return _PendingQKV(self, send, n)

    def ulysses_send_buffer(self, n, c, like, n_local):
        """(P, chunk, 3, C/P) all-to-all send buffer: block p = head group p of this rank's tokens, per token
        [q_g | k_g | v_g]; rows >= n_local (short trailing ranks) are zeroed.  The device path fills it straight from
        the RMSNorm+RoPE kernel (fg_rmsnorm_rope_grouped_bf16) and fg_copy_groups_bf16."""
        p, size = self.world_size, self.chunk(n)
        send = torch.empty((p, size, 3, c // p), dtype=like.dtype, device=like.device)
        if n_local < size:
            send[:, n_local:].zero_()
        return send

    def ulysses_exchange_async(self, send, n):
        """Start the all-to-all of a filled send buffer; `.wait()` as for ulysses_qkv_async."""
        return _PendingQKV(self, send, n)

    def ulysses_out_buffer(self, n, cols, like):
        """(P*chunk, cols) buffer for the head-group attention output: the kernel writes rows [0, N), the reverse
        all-to-all sends equal row blocks (rows >= N are padding of the trailing ranks)."""
        return torch.empty((self.world_size * self.chunk(n), cols), dtype=like.dtype, device=like.device)

    def ulysses_out_async(self, o_full, n, n_local):
        """o_full (P*chunk, (H/P)*D): this rank's head group for all tokens (rows >= N ignored).  Starts the reverse
        all-to-all; `.wait()` -> (1, n_local, H*D) for this rank's tokens, all heads."""
        return _PendingOut(self, o_full, n, n_local)

    def shared_seed(self):
        """A random seed drawn by group rank 0 and broadcast, so that the ranks working on ONE clip start from the same
        noise when the caller passes the reference's default seed=None."""
        t = torch.randint(0, 2 ** 31 - 1, (1,), dtype=torch.int64)
        if self.active:
            if dist.get_backend(self.group) != "gloo":
                t = t.to(torch.device("cuda", torch.cuda.current_device()))
            self.broadcast(t, 0)
        return int(t.item())

    # ------------------------------------------------------------------ VAE tiles
    def broadcast(self, tensor, src):
        """In-place broadcast from group rank `src` (VAE tiles decoded by different ranks)."""
        if self.active:
            gsrc = dist.get_global_rank(self.group, src) if self.group is not None else src
            if _staged(tensor, self.group):
                host = tensor.cpu()
                dist.broadcast(host, src=gsrc, group=self.group)
                tensor.copy_(host)
            else:
                dist.broadcast(tensor, src=gsrc, group=self.group)
        return tensor


class _PendingKV:
    def __init__(self, shard, k, v, n):
        self.n, self.works = n, []
        if not shard.active:
            self.kf, self.vf = k[0], v[0]
            return
        size = shard.chunk(n)
        if _staged(k, shard.group):      # single-GPU rehearsal backend: synchronous, through the host
            self.kf, self.vf = shard._gather_rows(k[0], size), shard._gather_rows(v[0], size)
            return
        bufs = []
        for t in (k[0], v[0]):
            buf = shard._padded(t, size)
            full = torch.empty((shard.world_size * size, t.shape[-1]), dtype=t.dtype, device=t.device)
            self.works.append(dist.all_gather_into_tensor(full, buf, group=shard.group, async_op=True))
            bufs.append((full, buf))
        self._keep = bufs      # keep the send buffers alive until wait()
        self.kf, self.vf = bufs[0][0], bufs[1][0]

    def wait(self):
        for w in self.works:
            w.wait()
        return self.kf[: self.n].unsqueeze(0), self.vf[: self.n].unsqueeze(0)


def _all_to_all_rows(shard, send, recv):
    """Equal-split all-to-all over dim 0; returns the async work (None when it already completed)."""
    if _staged(send, shard.group):
        hs, hr = send.cpu(), torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_to_all_single(hr, hs, group=shard.group)
        recv.copy_(hr)
        return None
    return dist.all_to_all_single(recv, send, group=shard.group, async_op=True)


class _PendingQKV:
    """Send layout (P, chunk, 3, g): block p = head group p of my tokens, per token [q_g | k_g | v_g] (g = H/P*D
    columns).  Receive layout (P, chunk, 3, g) = (P*chunk tokens in global order, 3g): q / k / v of my head group
    are column slices with row stride 3g — the attention kernel takes leading dimensions, so nothing is repacked."""

    def __init__(self, shard, send, n):
        p, size, _, g = send.shape
        assert (p, size) == (shard.world_size, shard.chunk(n)) and send.is_contiguous()
        self.n, self.g = n, g
        self.recv = torch.empty((p * size, 3 * g), dtype=send.dtype, device=send.device)
        self._send = send
        self.work = _all_to_all_rows(shard, send.view(p * size, 3 * g), self.recv)

    def wait(self):
        if self.work is not None:
            self.work.wait()
        r, g = self.recv[: self.n].unsqueeze(0), self.g
        return r[..., :g], r[..., g:2 * g], r[..., 2 * g:]


class _PendingOut:
    def __init__(self, shard, o_full, n, n_local):
        p, size = shard.world_size, shard.chunk(n)
        assert o_full.shape[0] == p * size and o_full.is_contiguous()
        self.p, self.size, self.n_local = p, size, n_local
        self._send = o_full
        self.recv = torch.empty_like(o_full)
        self.work = _all_to_all_rows(shard, o_full, self.recv)

    def wait_blocks(self):
        """-> (P, chunk, g): block p = head group p of this rank's tokens (rows >= n_local are padding)."""
        if self.work is not None:
            self.work.wait()
        return self.recv.view(self.p, self.size, self.recv.shape[-1])

    def wait(self):
        blocks = self.wait_blocks()[:, : self.n_local]                              # (head group, token, g)
        return blocks.transpose(0, 1).reshape(1, self.n_local, -1)                  # one copy: (token, all heads)


# ---------------------------------------------------------------------------------------------- rank layout
_subgroups = {}


def _subgroup(ranks):
    """Process group over `ranks` (global ranks), created once per rank set; every process must call this for every
    set in the same order (torch.distributed.new_group contract)."""
    key = tuple(ranks)
    if key not in _subgroups:
        _subgroups[key] = dist.new_group(list(ranks))
    return _subgroups[key]


class ParallelLayout:
    """world = cfg_parallel x sp.  cfg_parallel = 1: every rank runs both CFG branches on its token shard of the
    whole world.  cfg_parallel = 2: ranks [0, W/2) run the positive branch, [W/2, W) the negative one, each half
    sequence-parallel over its own group; one world all-gather per step exchanges the predictions."""

    def __init__(self, cfg_parallel=1, attn_mode="allgather", ranks=None):
        """ranks: the global ranks that work on ONE clip together (default: every rank of the default group).  Several
        disjoint rank sets = replicas serving different clips (replica_ranks()); every process must construct the layouts
        of ALL replicas in the same order (process-group creation is collective), see fairygen_amd.batch."""
        initialised = dist.is_available() and dist.is_initialized()
        everyone = list(range(dist.get_world_size())) if initialised else [0]
        ranks = everyone if ranks is None else list(ranks)
        member = (dist.get_rank() if initialised else 0) in ranks
        base = None if ranks == everyone else _subgroup(ranks)
        w = len(ranks)
        if cfg_parallel not in (1, 2) or w % cfg_parallel != 0:
            raise ValueError(f"cfg_parallel must be 1 or 2 and divide the group size {w}, got {cfg_parallel}")
        self.ranks, self.member = ranks, member
        self.cfg_parallel, self.attn_mode = (cfg_parallel if w > 1 else 1), attn_mode
        halves = None
        if self.cfg_parallel == 2:
            sp = w // 2
            halves = [_subgroup(ranks[b * sp:(b + 1) * sp]) for b in range(2)]        # both created on every process
        if not member:            # this process only took part in creating the groups
            self.world = self.shard = self.branch = None
            return
        self.world = TokenShard(base, attn_mode)
        if self.cfg_parallel == 1:
            self.branch, self.shard = None, self.world
        else:
            self.branch = self.world.rank // (w // 2)
            self.shard = TokenShard(halves[self.branch], attn_mode)

    @property
    def sp(self):
        return self.shard.world_size

    def describe(self):
        tag = f"cfg{self.cfg_parallel}xsp{self.sp}"
        return tag + (f"-{self.attn_mode}" if self.sp > 1 else "")

    def gather_branches(self, out_local, n):
        """out_local (1, n_local, C): this rank's token shard of ITS branch's prediction -> (2, N, C): the positive
        and the negative prediction over all tokens, on every rank (one world all-gather)."""
        assert self.cfg_parallel == 2
        size = self.shard.chunk(n)
        full = self.world._gather_rows(out_local[0], size)                      # (2*sp*size, C), branch-major
        return full.view(2, self.sp * size, -1)[:, :n]


def replica_ranks(world_size, replica_size):
    """Disjoint, contiguous rank sets of `replica_size` GPUs each (one clip per set at a time): [[0,1],[2,3],...]."""
    if replica_size < 1 or world_size % replica_size != 0:
        raise ValueError(f"replica_size {replica_size} must divide the world size {world_size}")
    return [list(range(r, r + replica_size)) for r in range(0, world_size, replica_size)]


def assign_tiles(costs, world):
    """VAE tile -> rank: longest-processing-time-first list scheduling (deterministic, same on every rank).
    Returns owner[i] for tile i.  6 tiles of unequal area at 704x1280 on 4 ranks: max load 2348 latent pixels
    instead of 3068 for round-robin."""
    load, owner = [0] * world, [0] * len(costs)
    for i in sorted(range(len(costs)), key=lambda i: (-costs[i], i)):
        r = min(range(world), key=lambda r: (load[r], r))
        owner[i] = r
        load[r] += costs[i]
    return owner

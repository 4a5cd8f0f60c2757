"""Latent-temporal token sharding of the DiT forward across the GPUs of one node (RCCL over xGMI).

Specification = the reference's USP path (``utils/xfuser/xdit_context_parallel.py:57-146`` and
``pipelines/wan_video.py:1224-1227,1310-1315,1379-1382``): tokens are frame-major, so a contiguous token
range IS a latent-temporal shard; every rank runs all 30 blocks on its N/P tokens, everything except
self-attention is token-local, and the head output is all-gathered at the end.  Instead of xfuser's four
Ulysses all-to-alls per layer we all-gather K and V (after RMSNorm+RoPE) once per layer: Wan's attention is
full 3-D, so the exact "halo" of a temporal shard is the whole sequence.  On the fully
connected xGMI mesh each peer's 42 MB (N=27 280, P=8) crosses its own link.

One process per GPU; the process group is torch.distributed's ("nccl" == RCCL on ROCm; "gloo" in CPU tests).
"""
import torch
import torch.distributed as dist


class TokenShard:
    def __init__(self, group=None):
        self.group = group
        if dist.is_available() and dist.is_initialized():
            self.world_size, self.rank = dist.get_world_size(group), dist.get_rank(group)
        else:
            self.world_size, self.rank = 1, 0

    def chunk(self, n):
        """Rows per rank: ceil split like torch.chunk (wan_video.py:1312); trailing ranks may be short/empty."""
        return (n + self.world_size - 1) // self.world_size

    def local_range(self, n, rank=None):
        rank = self.rank if rank is None else rank
        size = self.chunk(n)
        lo = min(rank * size, n)
        return lo, min(lo + size, n)

    def _gather_rows(self, local, size):
        """local (rows<=size, C) -> (world*size, C); short ranks are zero padded (only a suffix is padding)."""
        c = local.shape[-1]
        if local.shape[0] == size and local.is_contiguous():
            buf = local
        else:
            buf = torch.zeros((size, c), dtype=local.dtype, device=local.device)
            buf[: local.shape[0]].copy_(local)
        full = torch.empty((self.world_size * size, c), dtype=local.dtype, device=local.device)
        if local.is_cuda and dist.get_backend(self.group) == "gloo":
            # backend without device-tensor collectives (single-GPU rehearsal): stage through the host
            host = torch.empty(full.shape, dtype=full.dtype)
            dist.all_gather_into_tensor(host, buf.cpu(), group=self.group)
            full.copy_(host)
        else:
            dist.all_gather_into_tensor(full, buf, group=self.group)
        return full

    def all_gather_kv(self, k, v, n=None):
        """k, v (1, n_local, C) of this rank's tokens -> (1, N, C) of all tokens, in token order."""
        if self.world_size == 1:
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
        if self.world_size == 1:
            return x
        return self._gather_rows(x[0], self.chunk(n))[:n].unsqueeze(0).contiguous()

    def broadcast(self, tensor, src):
        """In-place broadcast from group rank `src` (VAE tiles decoded round-robin over ranks)."""
        if self.world_size > 1:
            gsrc = dist.get_global_rank(self.group, src) if self.group is not None else src
            if tensor.is_cuda and dist.get_backend(self.group) == "gloo":
                host = tensor.cpu()
                dist.broadcast(host, src=gsrc, group=self.group)
                tensor.copy_(host)
            else:
                dist.broadcast(tensor, src=gsrc, group=self.group)
        return tensor


class _PendingKV:
    def __init__(self, shard, k, v, n):
        self.n, self.works = n, []
        if shard.world_size == 1:
            self.kf, self.vf = k[0], v[0]
            return
        size = shard.chunk(n)
        staged = k.is_cuda and dist.get_backend(shard.group) == "gloo"
        if staged:      # single-GPU rehearsal backend: synchronous, through the host
            self.kf, self.vf = shard._gather_rows(k[0], size), shard._gather_rows(v[0], size)
            return
        bufs = []
        for t in (k[0], v[0]):
            if t.shape[0] == size and t.is_contiguous():
                buf = t
            else:
                buf = torch.zeros((size, t.shape[-1]), dtype=t.dtype, device=t.device)
                buf[: t.shape[0]].copy_(t)
            full = torch.empty((shard.world_size * size, t.shape[-1]), dtype=t.dtype, device=t.device)
            self.works.append(dist.all_gather_into_tensor(full, buf, group=shard.group, async_op=True))
            bufs.append((full, buf))
        self._keep = bufs      # keep the send buffers alive until wait()
        self.kf, self.vf = bufs[0][0], bufs[1][0]

    def wait(self):
        for w in self.works:
            w.wait()
        return self.kf[: self.n].unsqueeze(0), self.vf[: self.n].unsqueeze(0)

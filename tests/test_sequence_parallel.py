"""N>1 path on CPU: world_size-2 (and 3) gloo process groups exercise the token sharding and the K/V all-gather
exchange of fairygen_amd.sequence_parallel, with the CPU oracle's attention standing in for the HIP kernel
(checker only): sharded attention over gathered K/V == full attention, row for row."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import seeded


def _worker(rank, world, port, n_tokens, result_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fairygen_amd.sequence_parallel import TokenShard
        from oracle import wan_dit
        shard = TokenShard()
        assert (shard.world_size, shard.rank) == (world, rank)
        heads, c = 2, 256
        q, k, v = seeded((1, n_tokens, c), 1), seeded((1, n_tokens, c), 2), seeded((1, n_tokens, 3 * c), 3)[..., c:2 * c]
        lo, hi = shard.local_range(n_tokens)
        ranges = [shard.local_range(n_tokens, r) for r in range(world)]
        assert ranges[0][0] == 0 and ranges[-1][1] == n_tokens and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
        kf, vf = shard.all_gather_kv(k[:, lo:hi], v[:, lo:hi], n_tokens)           # v: strided slice, like the fused QKV buffer
        assert torch.equal(kf, k) and torch.equal(vf, v)
        kf2, _ = shard.all_gather_kv(k[:, lo:hi].contiguous(), v[:, lo:hi].contiguous())   # N inferred by all-reduce
        assert torch.equal(kf2, k)
        pend_a = shard.all_gather_kv_async(k[:, lo:hi], v[:, lo:hi], n_tokens)             # two gathers in flight, as the
        pend_b = shard.all_gather_kv_async(q[:, lo:hi], k[:, lo:hi], n_tokens)             # interleaved CFG branches do
        (ka, va), (kb, vb) = pend_a.wait(), pend_b.wait()
        assert torch.equal(ka, k) and torch.equal(va, v) and torch.equal(kb, q) and torch.equal(vb, k)
        out_local = wan_dit.attention(q[:, lo:hi], kf, vf, heads)
        full = shard.all_gather_tokens(out_local, n_tokens)
        want = wan_dit.attention(q, k, v, heads)
        # CPU SDPA blocks the query rows differently for a shard than for the full tensor: allow 1 bf16 ulp
        assert (full.float() - want.float()).abs().max().item() <= 2.0 ** -7 * want.float().abs().max().item(), \
            "sharded attention differs from the full one"
        tile = seeded((1, 3, 2, 4, 4), 9) if rank == 1 else torch.empty((1, 3, 2, 4, 4), dtype=torch.bfloat16)
        shard.broadcast(tile, src=1)
        assert torch.equal(tile, seeded((1, 3, 2, 4, 4), 9))
        open(os.path.join(result_dir, f"ok{rank}"), "w").close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_tokens,port", [(2, 105, 29631), (2, 64, 29632), (3, 10, 29633)])
def test_token_shard_gloo(tmp_path, world, n_tokens, port):
    mp.spawn(_worker, args=(world, port, n_tokens, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(world))


def test_single_process_shard_is_identity():
    from fairygen_amd.sequence_parallel import TokenShard
    s = TokenShard()
    k = seeded((1, 7, 8), 1)
    assert s.world_size == 1 and s.local_range(7) == (0, 7)
    assert s.all_gather_kv(k, k)[0] is k and s.all_gather_tokens(k, 7) is k


# ------------------------------------------------------------------------------------------------------------------
# GPU: the sharded DiT forward (HIP kernels + K/V all-gather) with 2 processes sharing the one MI355X of the test box.
# RCCL cannot put two ranks on one device, so the group is gloo; TokenShard stages device tensors through the host
# for backends without device support.  On the 8-GPU node the same code runs over "nccl" (= RCCL).
def _gpu_worker(rank, world, port, result_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fairygen_amd import synthetic
        from fairygen_amd.sequence_parallel import TokenShard
        from fairygen_amd.wan_video import model_fn_wan_video
        from fairygen_amd.wan_video_dit import WanModel
        cfg = dict(synthetic.TINY_DIT_KWARGS, dim=512, num_heads=4, ffn_dim=1024)
        sd = synthetic.random_state_dict(synthetic.dit_shapes(cfg), seed=77)
        m = WanModel(**cfg)
        m.load_state_dict(sd)
        m = m.to(device="cuda", dtype=torch.bfloat16).eval()
        lat = seeded((1, 48, 3, 10, 14), 5).cuda()          # 105 tokens: ragged split 53 + 52
        ctx = seeded((1, 24, 128), 6).cuda()
        ts = torch.tensor([500.0]).to(torch.bfloat16)
        with torch.no_grad():
            full = model_fn_wan_video(m, latents=lat, timestep=ts, context=ctx, fuse_vae_embedding_in_latents=True)
            shard = model_fn_wan_video(m, latents=lat, timestep=ts, context=ctx, fuse_vae_embedding_in_latents=True,
                                       sequence_shard=TokenShard())
        torch.cuda.synchronize()
        err = (full.float() - shard.float()).abs().max().item()
        assert err <= 2.0 ** -6 * full.float().abs().max().item(), f"sharded forward differs: {err}"
        # the denoise loop: sharded runs interleave the two CFG branches around their K/V gathers
        from fairygen_amd.wan_video import WanVideoPipeline
        ctx_n, z0 = seeded((1, 24, 128), 7).cuda(), seeded((1, 48, 1, 10, 14), 8).cuda()
        outs = []
        for sharded in (False, True):
            pipe = WanVideoPipeline(device="cuda", torch_dtype=torch.bfloat16)
            pipe.dit = m
            if sharded:
                pipe.enable_sequence_parallel()
            pipe.scheduler.set_timesteps(3, denoising_strength=1.0, shift=5.0)
            lat0 = lat.clone()
            lat0[:, :, 0:1] = z0
            shared = {"latents": lat0, "fuse_vae_embedding_in_latents": True, "first_frame_latents": z0}
            with torch.no_grad():
                outs.append(pipe.denoise(shared, {"context": ctx}, {"context": ctx_n}, 5.0, progress_bar_cmd=lambda x: x))
        torch.cuda.synchronize()
        err = (outs[0].float() - outs[1].float()).abs().max().item()
        assert err <= 0.05 * outs[0].float().abs().max().item(), f"sharded denoise loop differs: {err}"
        open(os.path.join(result_dir, f"ok{rank}"), "w").close()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_sharded_dit_forward_on_gpu(tmp_path):
    mp.spawn(_gpu_worker, args=(2, 29641, str(tmp_path)), nprocs=2, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(2))

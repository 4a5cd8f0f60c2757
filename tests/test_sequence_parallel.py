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


def _ulysses_worker(rank, world, port, n_tokens, heads, result_dir):
    """The Ulysses exchange: token shard -> head shard -> (oracle attention per head group) -> token shard must equal
    full attention on the unsharded tensors; two exchanges in flight like the interleaved CFG branches."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fairygen_amd.sequence_parallel import TokenShard
        from oracle import wan_dit
        shard = TokenShard(attn_mode="ulysses")
        d = 16
        c = heads * d
        q, k, qkv = seeded((1, n_tokens, c), 1), seeded((1, n_tokens, c), 2), seeded((1, n_tokens, 3 * c), 3)
        v = qkv[..., 2 * c:]                                    # strided slice, like the fused QKV buffer
        lo, hi = shard.local_range(n_tokens)
        hl = shard.heads_local(heads)
        g = hl * d
        pend_a = shard.ulysses_qkv_async(q[:, lo:hi], k[:, lo:hi], v[:, lo:hi], n_tokens, heads)
        pend_b = shard.ulysses_qkv_async(k[:, lo:hi], q[:, lo:hi], v[:, lo:hi], n_tokens, heads)
        qg, kg, vg = pend_a.wait()
        kb, qb, _ = pend_b.wait()
        cols = slice(rank * g, (rank + 1) * g)
        assert qg.shape == (1, n_tokens, g) and qg.stride(1) == 3 * g
        assert torch.equal(qg, q[..., cols]) and torch.equal(kg, k[..., cols]) and torch.equal(vg, v[..., cols])
        assert torch.equal(qb, q[..., cols]) and torch.equal(kb, k[..., cols])
        o_full = shard.ulysses_out_buffer(n_tokens, g, qg)
        o_full[:n_tokens] = wan_dit.attention(qg, kg, vg, hl)[0]
        out = shard.ulysses_out_async(o_full, n_tokens, hi - lo).wait()
        want = wan_dit.attention(q, k, v, heads)[:, lo:hi]
        assert out.shape == want.shape
        assert (out.float() - want.float()).abs().max().item() <= 2.0 ** -7 * want.float().abs().max().item()
        open(os.path.join(result_dir, f"ok{rank}"), "w").close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_tokens,heads,port", [(2, 105, 4, 29634), (3, 10, 6, 29635), (4, 64, 4, 29636),
                                                       (8, 43, 24, 29644)])      # the 8-GPU case: 3 of 24 heads per rank
def test_ulysses_exchange_gloo(tmp_path, world, n_tokens, heads, port):
    mp.spawn(_ulysses_worker, args=(world, port, n_tokens, heads, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(world))


def _layout_worker(rank, world, port, result_dir):
    """world = cfg_parallel x sp: group membership, branch assignment and the one-collective exchange of predictions."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fairygen_amd.sequence_parallel import ParallelLayout
        n, c = 21, 6                                            # ragged: chunk 11 -> 11 + 10
        preds = [seeded((1, n, c), 40), seeded((1, n, c), 41)]  # positive / negative prediction
        for mode in ("ulysses", "allgather"):
            lay = ParallelLayout(cfg_parallel=2, attn_mode=mode)
            sp = world // 2
            assert (lay.sp, lay.branch, lay.shard.rank) == (sp, rank // sp, rank % sp)
            assert lay.describe() == (f"cfg2xsp{sp}-{mode}" if sp > 1 else "cfg2xsp1")
            lo, hi = lay.shard.local_range(n)
            both = lay.gather_branches(preds[lay.branch][:, lo:hi], n)
            assert both.shape == (2, n, c) and torch.equal(both[0], preds[0][0]) and torch.equal(both[1], preds[1][0])
            t = torch.tensor([float(rank)])
            dist.all_reduce(t, group=lay.shard.group)           # the subgroup really is this rank's half
            assert t.item() == sum(range(lay.branch * sp, (lay.branch + 1) * sp))
        one = ParallelLayout(cfg_parallel=1, attn_mode="ulysses")
        assert one.branch is None and one.shard is one.world and one.describe() == f"cfg1xsp{world}-ulysses"
        with pytest.raises(ValueError):
            ParallelLayout(cfg_parallel=3)
        open(os.path.join(result_dir, f"ok{rank}"), "w").close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,port", [(2, 29637), (4, 29638), (8, 29645)])
def test_cfg_parallel_layout_gloo(tmp_path, world, port):
    mp.spawn(_layout_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(world))


def _replica_worker(rank, world, port, result_dir):
    """Multi-shot serving: disjoint replicas (2 ranks per clip, one CFG branch each); collectives never cross replicas."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fairygen_amd.batch import ShotScheduler
        sched = ShotScheduler(replica_size=2)
        assert sched.replicas == [[0, 1], [2, 3]] and sched.replica_id == rank // 2 and sched.is_writer == (rank % 2 == 0)
        lay = sched.layout
        assert (lay.cfg_parallel, lay.sp, lay.branch, lay.world.world_size, lay.world.rank) == (2, 1, rank % 2, 2, rank % 2)
        shots = [f"s{i}" for i in range(5)]
        assert sched.my_shots(shots) == shots[sched.replica_id::2]
        n, c = 9, 4
        mine = seeded((1, n, c), 100 + rank)                     # each rank's "prediction" (sp = 1: all tokens)
        both = lay.gather_branches(mine, n)                      # exchanged inside the replica only
        r0 = 2 * sched.replica_id
        assert torch.equal(both[0], seeded((1, n, c), 100 + r0)[0]) and torch.equal(both[1], seeded((1, n, c), 101 + r0)[0])
        tile = seeded((2, 3), 200 + rank)
        lay.world.broadcast(tile, src=1)                         # replica-local rank 1 = global rank r0 + 1
        assert torch.equal(tile, seeded((2, 3), 201 + r0))
        one = ShotScheduler(replica_size=4, cfg_parallel=1, attn_mode="allgather")
        assert one.n_replicas == 1 and one.layout.sp == 4 and one.layout.world.group is None
        open(os.path.join(result_dir, f"ok{rank}"), "w").close()
    finally:
        dist.destroy_process_group()


def test_shot_scheduler_replicas_gloo(tmp_path):
    mp.spawn(_replica_worker, args=(4, 29639, str(tmp_path)), nprocs=4, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(4))


def test_list_shots_and_single_process_scheduler(tmp_path):
    from fairygen_amd.batch import ShotScheduler, list_shots
    from fairygen_amd.sequence_parallel import replica_ranks
    for name in ("2.png", "10.png", "1.png", "1.txt", "10.txt", "notes.md"):
        (tmp_path / name).write_text("a pig walks")
    shots = list_shots(str(tmp_path))                            # "2.png" has no prompt: skipped, like the reference
    assert [s[0] for s in shots] == ["1", "10"]
    sched = ShotScheduler(replica_size=2)                        # no process group: one replica of one rank
    assert sched.n_replicas == 1 and sched.layout is None and sched.is_writer and sched.my_shots(shots) == shots
    assert replica_ranks(8, 2) == [[0, 1], [2, 3], [4, 5], [6, 7]]
    with pytest.raises(ValueError):
        replica_ranks(8, 3)


def _windows_worker(rank, world, port, result_dir):
    """attn_mode="windows": the sliding-window mode's windows dealt to ranks + one all-gather + the reference's blend must equal
    the sequential TemporalTiler result bit for bit (CPU tensors, a stand-in window function)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fairygen_amd.sequence_parallel import TokenShard
        from fairygen_amd.wan_video import temporal_tiler_steps, temporal_windows
        lat = seeded((1, 6, 11, 4, 5), 70)

        def window_fn(win):                      # a generator like model_fn_wan_video_steps, without exchanges
            return (win.float() * 1.5 + win.float().mean(dim=2, keepdim=True)).to(win.dtype)
            yield

        def run(shard):
            gen = temporal_tiler_steps(window_fn, lat, 4, 3, shard)
            while True:
                try:
                    next(gen)
                except StopIteration as done:
                    return done.value
        assert temporal_windows(11, 4, 3) == [(0, 4), (3, 7), (6, 10), (9, 11)]
        want = run(None)
        got = run(TokenShard(attn_mode="windows"))
        assert torch.equal(got, want)
        assert temporal_windows(7, 4, 2) == [(0, 4), (2, 6), (4, 7)] and temporal_windows(5, 8, 4) == [(0, 5)]
        open(os.path.join(result_dir, f"ok{rank}"), "w").close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,port", [(2, 29646), (3, 29647), (5, 29648)])
def test_window_parallel_tiler_gloo(tmp_path, world, port):
    mp.spawn(_windows_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(world))


def test_assign_tiles_balances_by_area():
    from fairygen_amd.sequence_parallel import assign_tiles
    from fairygen_amd.wan_video_vae import WanVideoVAE38
    tasks = WanVideoVAE38.tile_tasks(44, 80, (30, 52), (15, 26))                  # 704x1280: the reference's 6 tiles
    costs = [(min(h_, 44) - h) * (min(w_, 80) - w) for h, h_, w, w_ in tasks]
    assert costs == [1560, 1560, 840, 1508, 1508, 812]
    for world, bound in ((1, 7788), (2, 3908), (4, 2348), (8, 1560)):
        owner = assign_tiles(costs, world)
        loads = [sum(c for c, o in zip(costs, owner) if o == r) for r in range(world)]
        assert sum(loads) == sum(costs) and max(loads) <= bound, (world, loads)
    assert assign_tiles([], 4) == []


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
            for mode in ("allgather", "ulysses"):
                shard = model_fn_wan_video(m, latents=lat, timestep=ts, context=ctx, fuse_vae_embedding_in_latents=True,
                                           sequence_shard=TokenShard(attn_mode=mode))
                torch.cuda.synchronize()
                err = (full.float() - shard.float()).abs().max().item()
                assert err <= 2.0 ** -6 * full.float().abs().max().item(), f"sharded forward ({mode}) differs: {err}"
        # the denoise loop: sharded runs interleave the two CFG branches around their exchanges; cfg_parallel=2 gives
        # each half of the ranks one branch
        from fairygen_amd.wan_video import WanVideoPipeline
        ctx_n, z0 = seeded((1, 24, 128), 7).cuda(), seeded((1, 48, 1, 10, 14), 8).cuda()
        layouts = [None, (1, "allgather"), (1, "ulysses"), (2, "allgather"), (2, "ulysses")]
        outs = []
        for layout in layouts:
            pipe = WanVideoPipeline(device="cuda", torch_dtype=torch.bfloat16)
            pipe.dit = m
            if layout is not None:
                pipe.enable_sequence_parallel(cfg_parallel=layout[0], attn_mode=layout[1])
            pipe.scheduler.set_timesteps(3, denoising_strength=1.0, shift=5.0)
            lat0 = lat.clone()
            lat0[:, :, 0:1] = z0
            shared = {"latents": lat0, "fuse_vae_embedding_in_latents": True, "first_frame_latents": z0}
            with torch.no_grad():
                outs.append(pipe.denoise(shared, {"context": ctx}, {"context": ctx_n}, 5.0, progress_bar_cmd=lambda x: x))
        torch.cuda.synchronize()
        for layout, out in zip(layouts[1:], outs[1:]):
            err = (outs[0].float() - out.float()).abs().max().item()
            assert err <= 0.05 * outs[0].float().abs().max().item(), f"sharded denoise loop {layout} differs: {err}"
        open(os.path.join(result_dir, f"ok{rank}"), "w").close()
    finally:
        dist.destroy_process_group()


def _gpu_replica_worker(rank, world, port, result_dir):
    """4 gloo ranks on the one test GPU = 2 replicas x (one CFG branch per rank): each replica denoises its own clip
    (different seed) and must reproduce the single-process result of that clip."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fairygen_amd import synthetic
        from fairygen_amd.batch import ShotScheduler
        from fairygen_amd.wan_video import WanVideoPipeline
        from fairygen_amd.wan_video_dit import WanModel
        cfg = dict(synthetic.TINY_DIT_KWARGS, dim=512, num_heads=4, ffn_dim=1024)
        m = WanModel(**cfg)
        m.load_state_dict(synthetic.random_state_dict(synthetic.dit_shapes(cfg), seed=77))
        m = m.to(device="cuda", dtype=torch.bfloat16).eval()
        sched = ShotScheduler(replica_size=2)
        clip_seed = 300 + sched.replica_id
        lat, z0 = seeded((1, 48, 3, 10, 14), clip_seed).cuda(), seeded((1, 48, 1, 10, 14), 8).cuda()
        ctx, ctx_n = seeded((1, 24, 128), 6).cuda(), seeded((1, 24, 128), 7).cuda()
        outs = []
        for replicated in (False, True):
            pipe = WanVideoPipeline(device="cuda", torch_dtype=torch.bfloat16)
            pipe.dit = m
            if replicated:
                sched.attach(pipe)
                assert pipe.parallel.describe() == "cfg2xsp1"
            pipe.scheduler.set_timesteps(3, denoising_strength=1.0, shift=5.0)
            lat0 = lat.clone()
            lat0[:, :, 0:1] = z0
            shared = {"latents": lat0, "fuse_vae_embedding_in_latents": True, "first_frame_latents": z0}
            with torch.no_grad():
                outs.append(pipe.denoise(shared, {"context": ctx}, {"context": ctx_n}, 5.0, progress_bar_cmd=lambda x: x))
        torch.cuda.synchronize()
        assert torch.equal(outs[0], outs[1]), "a replica's clip must equal the single-process clip bit for bit"
        open(os.path.join(result_dir, f"ok{rank}"), "w").close()
    finally:
        dist.destroy_process_group()


def _gpu_windows_worker(rank, world, port, result_dir):
    """attn_mode="windows" on the device: the sliding-window denoise loop with its windows dealt over gloo ranks sharing the
    test GPU equals the single-process sliding-window loop bit for bit."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fairygen_amd import synthetic
        from fairygen_amd.wan_video import WanVideoPipeline
        from fairygen_amd.wan_video_dit import WanModel
        cfg = dict(synthetic.TINY_DIT_KWARGS, dim=512, num_heads=4, ffn_dim=1024)
        m = WanModel(**cfg)
        m.load_state_dict(synthetic.random_state_dict(synthetic.dit_shapes(cfg), seed=77))
        m = m.to(device="cuda", dtype=torch.bfloat16).eval()
        lat = seeded((1, 48, 9, 10, 14), 5).cuda()
        ctx, ctx_n = seeded((1, 24, 128), 6).cuda(), seeded((1, 24, 128), 7).cuda()
        outs = []
        for sharded in (False, True):
            pipe = WanVideoPipeline(device="cuda", torch_dtype=torch.bfloat16)
            pipe.dit = m
            if sharded:
                pipe.enable_sequence_parallel(attn_mode="windows")
            pipe.scheduler.set_timesteps(2, denoising_strength=1.0, shift=5.0)
            shared = {"latents": lat.clone(), "fuse_vae_embedding_in_latents": False, "first_frame_latents": None,
                      "sliding_window_size": 4, "sliding_window_stride": 2}
            with torch.no_grad():
                outs.append(pipe.denoise(shared, {"context": ctx}, {"context": ctx_n}, 5.0, progress_bar_cmd=lambda x: x))
        torch.cuda.synchronize()
        assert torch.equal(outs[0], outs[1])
        with pytest.raises(ValueError, match="sliding_window_size"):
            pipe.denoise({"latents": lat.clone(), "fuse_vae_embedding_in_latents": False}, {"context": ctx}, {"context": ctx_n}, 5.0,
                         progress_bar_cmd=lambda x: x)
        open(os.path.join(result_dir, f"ok{rank}"), "w").close()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_window_parallel_on_gpu(tmp_path):
    mp.spawn(_gpu_windows_worker, args=(3, 29649, str(tmp_path)), nprocs=3, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(3))


@pytest.mark.gpu
def test_replicas_on_gpu(tmp_path):
    mp.spawn(_gpu_replica_worker, args=(4, 29643, str(tmp_path)), nprocs=4, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(4))


@pytest.mark.gpu
@pytest.mark.parametrize("world,port", [(2, 29641), (4, 29642)])
def test_sharded_dit_forward_on_gpu(tmp_path, world, port):
    mp.spawn(_gpu_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(world))


# ------------------------------------------------------------------------------------------------------------------
# GPU + RCCL: the DEVICE-collective branches (all_gather_into_tensor / all_to_all_single(async_op=True) / broadcast /
# all_reduce on HIP tensors, the interleaved generators with two exchanges in flight, barrier(device_ids=)).  On the one-GPU
# test box they run on a 1-rank "nccl" (= RCCL) communicator with FAIRYGEN_FORCE_COLLECTIVES=1, which turns off every
# "nothing to exchange" short-cut of a 1-rank group; with >= 2 GPUs visible the same worker runs one rank per device.
def _rccl_worker(rank, world, port, result_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0", FAIRYGEN_FORCE_COLLECTIVES="1")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)      # before any other GPU work
    try:
        from fairygen_amd import hip, synthetic
        from fairygen_amd.sequence_parallel import ParallelLayout, TokenShard
        from fairygen_amd.wan_video import WanVideoPipeline, model_fn_wan_video
        from fairygen_amd.wan_video_dit import WanModel
        from fairygen_amd.wan_video_vae import WanVideoVAE38
        assert dist.get_backend() == "nccl"
        exact = world == 1                      # one rank: same kernels on the same rows -> bit-identical
        dist.barrier(device_ids=[rank])         # bench.py's barrier form
        t = torch.tensor([float(rank + 1)], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)      # bench.py's sync_max
        assert t.item() == world

        # --- exchange primitives on device tensors
        shard = TokenShard()
        assert shard.active and (shard.world_size, shard.rank) == (world, rank)
        heads, c, n = 4, 512, 203
        q, k, qkv = seeded((1, n, c), 1).to(dev), seeded((1, n, c), 2).to(dev), seeded((1, n, 3 * c), 3).to(dev)
        v = qkv[..., c:2 * c]
        lo, hi = shard.local_range(n)
        kf, vf = shard.all_gather_kv(k[:, lo:hi], v[:, lo:hi], n)
        assert torch.equal(kf, k) and torch.equal(vf, v)
        kf2, _ = shard.all_gather_kv(k[:, lo:hi].contiguous(), v[:, lo:hi].contiguous())      # N by a device all-reduce
        assert torch.equal(kf2, k)
        pa = shard.all_gather_kv_async(k[:, lo:hi], v[:, lo:hi], n)      # two async gathers in flight
        pb = shard.all_gather_kv_async(q[:, lo:hi], k[:, lo:hi], n)
        (ka, va), (kb, vb) = pa.wait(), pb.wait()
        assert torch.equal(ka, k) and torch.equal(va, v) and torch.equal(kb, q) and torch.equal(vb, k)
        want = hip.attention(q, k, v, heads)
        got = shard.all_gather_tokens(hip.attention(q[:, lo:hi].contiguous(), kf, vf, heads), n)
        assert torch.equal(got, want) if exact else (got.float() - want.float()).abs().max().item() <= 2.0 ** -7 * want.float().abs().max().item()
        us = TokenShard(attn_mode="ulysses")
        hl = us.heads_local(heads)
        g = hl * (c // heads)
        p1 = us.ulysses_qkv_async(q[:, lo:hi], k[:, lo:hi], v[:, lo:hi], n, heads)      # two all-to-alls in flight
        p2 = us.ulysses_qkv_async(k[:, lo:hi], q[:, lo:hi], v[:, lo:hi], n, heads)
        qg, kg, vg = p1.wait()
        kb2, qb2, _ = p2.wait()
        cols = slice(rank * g, (rank + 1) * g)
        assert torch.equal(qg, q[..., cols]) and torch.equal(kg, k[..., cols]) and torch.equal(vg, v[..., cols])
        assert torch.equal(qb2, q[..., cols]) and torch.equal(kb2, k[..., cols])
        o_full = us.ulysses_out_buffer(n, g, qg)
        hip.attention(qg, kg, vg, hl, out=o_full[:n].unsqueeze(0))
        out = us.ulysses_out_async(o_full, n, hi - lo).wait()
        assert torch.equal(out, want[:, lo:hi]) if exact else \
            (out.float() - want[:, lo:hi].float()).abs().max().item() <= 2.0 ** -7 * want.float().abs().max().item()
        tile = seeded((1, 3, 2, 4, 4), 9).to(dev) if rank == world - 1 else torch.empty((1, 3, 2, 4, 4), dtype=torch.bfloat16, device=dev)
        shard.broadcast(tile, src=world - 1)
        assert torch.equal(tile.cpu(), seeded((1, 3, 2, 4, 4), 9))
        seeds = torch.tensor([shard.shared_seed()], device=dev)
        lo_s, hi_s = seeds.clone(), seeds.clone()
        dist.all_reduce(lo_s, op=dist.ReduceOp.MIN), dist.all_reduce(hi_s, op=dist.ReduceOp.MAX)
        assert lo_s.item() == hi_s.item(), "ranks of one layout must agree on the seed"

        # --- the sharded DiT forward through the device collectives == the unsharded forward
        cfg = dict(synthetic.TINY_DIT_KWARGS, dim=512, num_heads=4, ffn_dim=1024)
        m = WanModel(**cfg)
        m.load_state_dict(synthetic.random_state_dict(synthetic.dit_shapes(cfg), seed=77))
        m = m.to(device=dev, dtype=torch.bfloat16).eval()
        lat = seeded((1, 48, 3, 10, 14), 5).to(dev)          # 105 tokens
        ctx, ctx_n = seeded((1, 24, 128), 6).to(dev), seeded((1, 24, 128), 7).to(dev)
        z0 = seeded((1, 48, 1, 10, 14), 8).to(dev)
        ts = torch.tensor([500.0]).to(torch.bfloat16)
        with torch.no_grad():
            full = model_fn_wan_video(m, latents=lat, timestep=ts, context=ctx, fuse_vae_embedding_in_latents=True)
            for mode in ("allgather", "ulysses"):
                got = model_fn_wan_video(m, latents=lat, timestep=ts, context=ctx, fuse_vae_embedding_in_latents=True,
                                         sequence_shard=TokenShard(attn_mode=mode))
                torch.cuda.synchronize()
                err = (full.float() - got.float()).abs().max().item()
                assert (torch.equal(full, got) if exact else err <= 2.0 ** -6 * full.float().abs().max().item()), \
                    f"sharded forward ({mode}) over RCCL differs: {err}"
        # --- the denoise loop: interleaved CFG branches (two exchanges in flight per layer); cfg_parallel=2 needs >= 2 ranks
        layouts = [None, (1, "allgather"), (1, "ulysses")] + ([(2, "allgather"), (2, "ulysses")] if world % 2 == 0 else [])
        outs = []
        for layout in layouts:
            pipe = WanVideoPipeline(device=dev, torch_dtype=torch.bfloat16)
            pipe.dit = m
            if layout is not None:
                pipe.enable_sequence_parallel(cfg_parallel=layout[0], attn_mode=layout[1])
                assert pipe.sequence_shard.active
            pipe.scheduler.set_timesteps(3, denoising_strength=1.0, shift=5.0)
            lat0 = lat.clone()
            lat0[:, :, 0:1] = z0
            shared = {"latents": lat0, "fuse_vae_embedding_in_latents": True, "first_frame_latents": z0}
            with torch.no_grad():
                outs.append(pipe.denoise(shared, {"context": ctx}, {"context": ctx_n}, 5.0, progress_bar_cmd=lambda x: x))
        torch.cuda.synchronize()
        for layout, o in zip(layouts[1:], outs[1:]):
            err = (outs[0].float() - o.float()).abs().max().item()
            assert (torch.equal(outs[0], o) if exact else err <= 0.05 * outs[0].float().abs().max().item()), \
                f"sharded denoise loop {layout} over RCCL differs: {err}"
        # --- sliding-window mode with the windows all-gathered on the device
        lat9 = seeded((1, 48, 9, 10, 14), 5).to(dev)
        wouts = []
        for sharded in (False, True):
            pipe = WanVideoPipeline(device=dev, torch_dtype=torch.bfloat16)
            pipe.dit = m
            if sharded:
                pipe.enable_sequence_parallel(attn_mode="windows")
            pipe.scheduler.set_timesteps(2, denoising_strength=1.0, shift=5.0)
            shared = {"latents": lat9.clone(), "fuse_vae_embedding_in_latents": False, "first_frame_latents": None,
                      "sliding_window_size": 4, "sliding_window_stride": 2}
            with torch.no_grad():
                wouts.append(pipe.denoise(shared, {"context": ctx}, {"context": ctx_n}, 5.0, progress_bar_cmd=lambda x: x))
        assert torch.equal(wouts[0], wouts[1])
        # --- tiled VAE decode: tiles dealt to the ranks and broadcast on the device == the single-process decode
        vae = WanVideoVAE38(dim=32, dec_dim=32)
        vae.load_state_dict(synthetic.random_state_dict(synthetic.vae_shapes(dec_dim=32, dim=32), seed=1234))
        vae = vae.to(device=dev, dtype=torch.bfloat16).eval()
        z = seeded((1, 48, 2, 8, 10), 11).to(dev)
        lay = ParallelLayout(cfg_parallel=1, attn_mode="allgather")
        with torch.no_grad():
            one = vae.decode(z, device=dev, tiled=True, tile_size=(6, 6), tile_stride=(3, 4))
            many = vae.decode(z, device=dev, tiled=True, tile_size=(6, 6), tile_stride=(3, 4), shard=lay.world)
        torch.cuda.synchronize()
        assert torch.equal(one, many), "tile-parallel decode over the device broadcast differs"
        dist.barrier(device_ids=[rank])
        open(os.path.join(result_dir, f"ok{rank}"), "w").close()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_rccl_device_collectives_one_rank(tmp_path):
    """Every device-collective branch of sequence_parallel.py on a 1-rank RCCL communicator, bit-equal to the unsharded path."""
    mp.spawn(_rccl_worker, args=(1, 29650, str(tmp_path)), nprocs=1, join=True)
    assert os.path.exists(tmp_path / "ok0")


@pytest.mark.gpu
def test_rccl_device_collectives_two_ranks(tmp_path):
    """The same worker with one rank per device over xGMI — runs wherever >= 2 GPUs are leased (skipped on the 1-GPU box)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL cannot put two ranks on one device)")
    mp.spawn(_rccl_worker, args=(2, 29651, str(tmp_path)), nprocs=2, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(2))


def test_forced_collectives_one_rank_gloo(tmp_path):
    """CPU twin of the 1-rank RCCL test: FAIRYGEN_FORCE_COLLECTIVES=1 makes a 1-rank group run its collectives."""
    mp.spawn(_forced_gloo_worker, args=(29652, str(tmp_path)), nprocs=1, join=True)
    assert os.path.exists(tmp_path / "ok0")


def _forced_gloo_worker(rank, port, result_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", FAIRYGEN_FORCE_COLLECTIVES="1")
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        from fairygen_amd.sequence_parallel import TokenShard
        s = TokenShard(attn_mode="ulysses")
        assert s.active and s.world_size == 1
        k, v = seeded((1, 7, 8), 1), seeded((1, 7, 8), 2)
        kf, vf = s.all_gather_kv(k, v, 7)
        assert kf is not k and torch.equal(kf, k) and torch.equal(vf, v)          # a real collective made a copy
        q2, k2, v2 = s.ulysses_qkv_async(k, v, k, 7, 2).wait()
        assert torch.equal(q2, k) and torch.equal(k2, v) and torch.equal(v2, k)
        assert isinstance(s.shared_seed(), int)
        open(os.path.join(result_dir, "ok0"), "w").close()
    finally:
        dist.destroy_process_group()

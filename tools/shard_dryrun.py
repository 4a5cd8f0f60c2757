"""Per-rank compute of every multi-GPU layout, measured on ONE GPU: the collectives are replaced by local copies of the
same size (no xGMI), so a step's time is what rank 0 of a P-rank job computes plus its launch overhead — the
communication-free upper bound of the scaling the driver's 8-GPU run can reach, and the place to see which layout
loses how much to small shapes.

    python tools/shard_dryrun.py [--steps 2] [--worlds 2,4,8]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from fairygen_amd import sequence_parallel as sp  # noqa: E402


class DryShard(sp.TokenShard):
    def __init__(self, world_size, rank, attn_mode):
        self.group, self.attn_mode, self.world_size, self.rank = None, attn_mode, world_size, rank
        self.active = world_size > 1

    def _gather_rows(self, local, size):
        # every "peer" contributes a copy of the local rows: same statistics as real data (all-zero peers would let the
        # MFMA kernels run cooler and faster than they do in a real job)
        full = torch.empty((self.world_size, size, local.shape[-1]), dtype=local.dtype, device=local.device)
        full[:, : local.shape[0]] = local
        if local.shape[0] < size:
            full[:, local.shape[0]:] = 0
        return full.view(self.world_size * size, -1)

    def broadcast(self, tensor, src):
        return tensor


class DryLayout:
    def __init__(self, world, cfg_parallel, attn_mode):
        self.cfg_parallel, self.attn_mode = cfg_parallel, attn_mode
        self.world = DryShard(world, 0, attn_mode)
        self.branch = 0 if cfg_parallel == 2 else None
        self.shard = DryShard(world // cfg_parallel, 0, attn_mode)

    describe = sp.ParallelLayout.describe
    sp = property(lambda self: self.shard.world_size)

    def gather_branches(self, out_local, n):
        size = self.shard.chunk(n)
        full = self.world._gather_rows(out_local[0], size)
        return full.view(2, self.sp * size, -1)[:, :n]


def _copy_rows(shard, send, recv):
    recv.copy_(send)
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--worlds", default="2,4,8")
    ap.add_argument("--only", default="", help="substring filter on the layout name (profiling one layout)")
    ap.add_argument("--no-vae", action="store_true")
    ap.add_argument("--height", type=int, default=704)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--frames", type=int, default=121)
    a = ap.parse_args()
    args = argparse.Namespace(layers=0, no_lora=True, height=a.height, width=a.width, frames=a.frames)
    device = "cuda:0"
    sp._staged = lambda t, group: True                       # take the synchronous branch of the pending objects ...
    sp._all_to_all_rows = _copy_rows                         # ... and copy instead of exchanging
    pipe, cfg = bench.build_pipeline(args, device)
    lat_shape = (1, 48, (a.frames - 1) // 4 + 1, a.height // 16, a.width // 16)
    noise = bench.seeded(lat_shape, 1).to(device)
    ctx_p, ctx_n = bench.seeded((1, 512, 4096), 2).to(device), bench.seeded((1, 512, 4096), 3).to(device)
    z0 = bench.seeded((1, 48, 1, lat_shape[3], lat_shape[4]), 4).to(device)

    def run(steps):
        pipe.scheduler.set_timesteps(steps, denoising_strength=1.0, shift=5.0)
        lat = noise.clone()
        shared = {"latents": lat, "fuse_vae_embedding_in_latents": True, "first_frame_latents": z0}
        return pipe.denoise(shared, {"context": ctx_p}, {"context": ctx_n}, 5.0, progress_bar_cmd=lambda x: x)

    def timed(steps):
        run(1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(steps)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps

    with torch.no_grad():
        base = timed(a.steps)
        print(f"single: {base * 1e3:.1f} ms/step")
        for world in [int(w) for w in a.worlds.split(",")]:
            for cfgp, mode in bench.candidate_layouts(world, cfg["num_heads"]):
                if a.only and a.only not in DryLayout(world, cfgp, mode).describe():
                    continue
                pipe.parallel = DryLayout(world, cfgp, mode)
                pipe.sequence_shard = pipe.parallel.shard
                t = timed(a.steps)
                print(f"world {world} {pipe.parallel.describe()}: {t * 1e3:.1f} ms/step per rank, no comm -> "
                      f"x{base / t:.2f} of {world} ({base / t / world * 100:.0f}%)", flush=True)
        if a.no_vae:
            return
        # VAE: the largest tile alone is the decode critical path at >= 6 ranks
        pipe.parallel = None
        lat = noise
        pipe.vae.decode(lat[:, :, :2, :8, :8].contiguous(), device=device, tiled=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pipe.vae.model.decode(lat[:, :, :, :30, :52].contiguous(), pipe.vae.scale)
        torch.cuda.synchronize()
        print(f"VAE largest tile (30x52 latent): {time.perf_counter() - t0:.2f} s")


if __name__ == "__main__":
    main()

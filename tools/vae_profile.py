"""Per-shape time of every VAE kernel call in the decode of one (30,52) latent tile (the largest of the reference's 6 tiles
at 704x1280): HIP events around each fairygen_amd.hip call, aggregated by (op, shape).

    python tools/vae_profile.py [--h 30 --w 52 --frames 31]
"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from fairygen_amd import hip  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--h", type=int, default=30)
    ap.add_argument("--w", type=int, default=52)
    ap.add_argument("--frames", type=int, default=31)
    ap.add_argument("--group", type=int, default=0, help="latent frames per decoder call after the first (0 = the model's default)")
    ap.add_argument("--top", type=int, default=1000, help="rows of the per-shape table to print")
    a = ap.parse_args()
    args = argparse.Namespace(layers=2, no_lora=True, height=704, width=1280, frames=121)
    pipe, _ = bench.build_pipeline(args, "cuda:0")
    z = bench.seeded((1, 48, a.frames, a.h, a.w), 1).to("cuda:0")
    records = []

    def wrap(name, shape_of):
        orig = getattr(hip, name)

        def timed(*args, **kw):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = orig(*args, **kw)
            e.record()
            records.append((name, shape_of(*args, **kw), s, e))
            return r
        setattr(hip, name, timed)

    wrap("conv3d_cl", lambda x, wp, b, cout, kt, ks, **kw: (tuple(x.shape), cout, kt, ks,
                                                          "up" if kw.get("upsample2x") else "", "interleave" if kw.get("time_interleave") else ""))
    wrap("vae_rmsnorm_silu", lambda x, g, silu=True, out=None: (tuple(x.shape),))
    wrap("dupup3d_add", lambda x, main, *r, **kw: (tuple(x.shape), tuple(main.shape)))
    wrap("vae_unpatchify", lambda x, *r, **kw: (tuple(x.shape),))
    if a.group:
        pipe.vae.model.max_chunk_group = a.group
    with torch.no_grad():
        pipe.vae.model.decode(z[:, :, :2, :8, :8].contiguous(), pipe.vae.scale)      # warm-up / weight packing
        records.clear()
        torch.cuda.synchronize()
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
        t0.record()
        pipe.vae.model.decode(z, pipe.vae.scale)
        t1.record()
        torch.cuda.synchronize()
    total = t0.elapsed_time(t1)
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for name, shape, s, e in records:
        ms = s.elapsed_time(e)
        fl = 0.0
        if name == "conv3d_cl":
            (t, h, w, cin), cout, kt, ks = shape[0], shape[1], shape[2], shape[3]
            t_out = t - (kt - 1)
            up = 2 if shape[4] else 1
            fl = 2.0 * t_out * h * up * w * up * cout * cin * kt * ks * ks
        k = (name, shape)
        agg[k][0] += 1; agg[k][1] += ms; agg[k][2] += fl
    print(f"decode of a (1,48,{a.frames},{a.h},{a.w}) tile: {total:.1f} ms; timed calls {sum(v[1] for v in agg.values()):.1f} ms")
    for (name, shape), (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:a.top]:
        tf = f"{fl / ms / 1e9:7.0f} TF/s" if fl else ""
        print(f"{ms:8.2f} ms {ms / total * 100:5.1f}%  x{n:4d}  {name:18s} {shape} {tf}")


if __name__ == "__main__":
    main()

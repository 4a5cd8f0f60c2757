"""Kernel micro-benchmarks on the GPU box (HIP events on the launch stream, random data).

    python tools/microbench.py attn --nq 27280 --nkv 27280 --heads 24
    python tools/microbench.py conv --cin 1024 --cout 1024 --t 1 --h 44 --w 80 --kt 3 --ks 3
    python tools/microbench.py elementwise --rows 27280
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fairygen_amd import hip  # noqa: E402


def timeit(fn, iters, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record()
        evs.append((s, e))
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in evs)
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["attn", "conv", "elementwise"])
    ap.add_argument("--nq", type=int, default=27280)
    ap.add_argument("--nkv", type=int, default=27280)
    ap.add_argument("--heads", type=int, default=24)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--cin", type=int, default=1024)
    ap.add_argument("--cout", type=int, default=1024)
    ap.add_argument("--t", type=int, default=1)
    ap.add_argument("--h", type=int, default=44)
    ap.add_argument("--w", type=int, default=80)
    ap.add_argument("--kt", type=int, default=3)
    ap.add_argument("--ks", type=int, default=3)
    ap.add_argument("--rows", type=int, default=27280)
    ap.add_argument("--interleaved", action="store_true")
    ap.add_argument("--per-head", action="store_true")
    ap.add_argument("--attn-scale", default="pow2", choices=["pow2", "model", "both"],
                    help="pow2: the scale the pipeline passes for self-attention (2^-3 / log2 e: the kernel's pre-multiplied form); model: 1/sqrt(d) (plain form)")
    a = ap.parse_args()
    hip.load()
    dev = "cuda"
    g = torch.Generator(dev).manual_seed(0)
    rnd = lambda *s: torch.randn(s, generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)  # noqa: E731
    if a.what == "attn":
        c = a.heads * 128
        if a.interleaved:      # q | k | v column slices of one (N, 3c) buffer, as after the Ulysses all-to-all
            assert a.nq == a.nkv
            qkv = rnd(1, a.nq, 3 * c)
            q, k, v = qkv[..., :c], qkv[..., c:2 * c], qkv[..., 2 * c:]
        else:
            q, k, v = rnd(1, a.nq, c), rnd(1, a.nkv, c), rnd(1, a.nkv, c)
        if a.per_head:         # (H, N, 128): every head's rows contiguous (row stride 256 B instead of H*256 B)
            q, k, v = (t.view(a.nq if t is q else a.nkv, a.heads, 128).permute(1, 0, 2).contiguous() for t in (q, k, v))
            out = torch.empty((a.heads, a.nq, 128), dtype=torch.bfloat16, device=dev)
            med, mn = timeit(lambda: hip.attention(q, k, v, 1, out=out), a.iters)
            fl = 4.0 * a.nq * a.nkv * c
            print(f"attn per-head layout nq={a.nq} nkv={a.nkv} H={a.heads}: median {med:.3f} ms ({fl / med / 1e9:.1f} TFLOP/s), min {mn:.3f} ms ({fl / mn / 1e9:.1f})")
            return
        out = torch.empty((1, a.nq, c), dtype=torch.bfloat16, device=dev)
        fl = 4.0 * a.nq * a.nkv * c
        for form in (("pow2", "model") if a.attn_scale == "both" else (a.attn_scale,)):
            scale = hip.pow2_softmax_scale(128)[0] if form == "pow2" else None
            med, mn = timeit(lambda: hip.attention(q, k, v, a.heads, out=out, scale=scale), a.iters)
            print(f"attn ({form} scale) nq={a.nq} nkv={a.nkv} H={a.heads}: median {med:.3f} ms ({fl / med / 1e9:.1f} TFLOP/s), min {mn:.3f} ms ({fl / mn / 1e9:.1f})")
    elif a.what == "conv":
        x = rnd(a.t + a.kt - 1, a.h, a.w, a.cin)
        w = rnd(a.cout, a.cin, a.kt, a.ks, a.ks) * (a.cin * a.kt * a.ks * a.ks) ** -0.5
        b = rnd(a.cout)
        wp = hip.conv_pack_weight(w)
        out = torch.empty((a.t, a.h, a.w, a.cout), dtype=torch.bfloat16, device=dev)
        med, mn = timeit(lambda: hip.conv3d_cl(x, wp, b, a.cout, a.kt, a.ks, out=out), a.iters)
        fl = 2.0 * a.t * a.h * a.w * a.cout * a.cin * a.kt * a.ks * a.ks
        print(f"conv {a.cin}->{a.cout} k=({a.kt},{a.ks},{a.ks}) @({a.t},{a.h},{a.w}): median {med:.3f} ms ({fl / med / 1e9:.1f} TFLOP/s), min {mn:.3f} ms")
    else:
        c = 3072
        x, y = rnd(1, a.rows, c), rnd(1, a.rows, c)
        mod = hip.ModTable(rnd(2, 6, c), 880)
        o1, o2 = torch.empty_like(x), torch.empty_like(x)
        by = a.rows * c * 2
        cos = torch.rand((a.rows, 64), device=dev, dtype=torch.float64)
        sin = (1 - cos * cos).sqrt()
        cs = torch.stack([cos, sin], dim=-1).float().contiguous()
        wide = rnd(1, a.rows, 3 * c)
        for name, fn, nbytes in [
            ("rmsnorm+rope fp64 tables (q slice of qkv)", lambda: hip.rmsnorm_rope(wide[..., :c], mod.table[0, 0], 24, 1e-6, cos, sin, out=o1), 2 * by),
            ("rmsnorm+rope fp32 table (q slice of qkv)", lambda: hip.rmsnorm_rope(wide[..., :c], mod.table[0, 0], 24, 1e-6, cs, out=o1), 2 * by),
            ("ln_modulate", lambda: hip.ln_modulate(x, mod, 0, 1, 1e-6, out=o1), 2 * by),
            ("residual_ln_modulate", lambda: hip.residual_ln_modulate(x, y, mod, 5, 0, 1, 1e-6, x_out=o1, norm_out=o2), 4 * by),
            ("rmsnorm(no rope)", lambda: hip.rmsnorm_rope(x, mod.table[0, 0], 24, 1e-6, out=o1), 2 * by),
            ("gelu", lambda: hip.activation(x, "gelu_tanh", out=o1), 2 * by),
        ]:
            med, mn = timeit(fn, a.iters)
            print(f"{name}: median {med * 1e3:.1f} us, {nbytes / med / 1e6:.0f} GB/s algorithmic")


if __name__ == "__main__":
    main()

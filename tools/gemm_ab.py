"""fg_gemm_bias_bf16 (hand-scheduled MFMA kernel) against F.linear (hipBLASLt) on the DiT shapes: correctness + interleaved timing.

    python tools/gemm_ab.py [--m 27280] [--stamp-lib path]      (--stamp-lib: a build of gen_gemm_w4.py --stamp: decode cycle stamps)
"""
import argparse
import ctypes
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fairygen_amd import hip  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=27280)
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--stamp-lib", default="")
    a = ap.parse_args()
    hip.load()
    dev = "cuda"
    g = torch.Generator(dev).manual_seed(0)
    rnd = lambda *s, sc=1.0: (torch.randn(s, generator=g, device=dev, dtype=torch.float32) * sc).to(torch.bfloat16)  # noqa: E731
    shapes = [("qkv", 3072, 9216), ("o", 3072, 3072), ("ffn.0", 3072, 14336), ("ffn.2", 14336, 3072)]
    stamp = None
    if a.stamp_lib:
        stamp = ctypes.CDLL(os.path.abspath(a.stamp_lib))
        stamp.fg_gemm_bias_bf16.restype = ctypes.c_int
        stamp.fg_gemm_bias_bf16.argtypes = hip._SIGNATURES["fg_gemm_bias_bf16"]
    for name, k, n in shapes:
        x, w, b = rnd(a.m, k), rnd(n, k, sc=0.02), rnd(n, sc=0.1)
        ref = F.linear(x, w, b)
        out = hip.gemm_bias(x, w, b)
        torch.cuda.synchronize()
        rows = torch.randint(0, a.m, (64,), device=dev)
        ref32 = x[rows].float() @ w.float().t() + b.float()
        e_ref = (ref[rows].float() - ref32).abs().max().item()
        e_out = (out[rows].float() - ref32).abs().max().item()
        same = (out == ref).float().mean().item()
        print(f"{name}: max|lib - f32| {e_ref:.4f}, max|w4 - f32| {e_out:.4f}, bit-identical to the library: {same:.4f}, last row ok: "
              f"{torch.equal(out[-1], ref[-1]) or (out[-1].float() - ref[-1].float()).abs().max().item()}", flush=True)
        fns = {"hipBLASLt": lambda: F.linear(x, w, b), "w4": lambda: hip.gemm_bias(x, w, b, out=out)}
        times = {kk: [] for kk in fns}
        for _ in range(2):
            for f in fns.values():
                f()
        torch.cuda.synchronize()
        for _ in range(a.rounds):
            for kk, f in fns.items():
                evs = []
                for _ in range(3):
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record(); f(); e.record()
                    evs.append((s, e))
                torch.cuda.synchronize()
                times[kk] += [s.elapsed_time(e) for s, e in evs]
        fl = 2.0 * a.m * k * n
        for kk, ts in times.items():
            ts = sorted(ts)
            print(f"   {kk}: median {ts[len(ts) // 2]:.3f} ms = {fl / ts[len(ts) // 2] / 1e9:.1f} TFLOP/s (min {ts[0]:.3f} ms)", flush=True)
        if stamp is not None:
            o2 = torch.zeros_like(out)
            st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            for _ in range(3):
                stamp.fg_gemm_bias_bf16(x.data_ptr(), k, w.data_ptr(), b.data_ptr(), o2.data_ptr(), n, a.m, n, k, 0, st)
            torch.cuda.synchronize()
            raw = o2[0::256, :].contiguous().view(torch.int32).view(-1, n // 256, 128)[:, :, :4].reshape(-1, 4).cpu().double()
            raw = raw[raw[:, 2] > 0]
            per = (raw[:, 0] / raw[:, 2]).median().item()
            clk = (raw[:, 0] / raw[:, 1] * 100).median().item()
            tot = raw[:, 3].median().item()
            print(f"   stamp: {len(raw)} tiles, {per:.0f} cycles per 64-MFMA step, clock {clk:.0f} MHz, loop {raw[:, 0].median().item():.0f} of {tot:.0f} cycles per tile", flush=True)


if __name__ == "__main__":
    main()

"""The hand-scheduled DiT GEMMs against F.linear (hipBLASLt) on the DiT shapes: correctness + interleaved timing.

    python tools/gemm_ab.py [--m 27280] [--only p|fp8] [--stamp-lib-p path]

  p  : fg_gemm_epilogue_bf16, the persistent form (one workgroup per CU, tile list, 256x256 tiles), mode 0, its residual modes
       (2: x + gate*y, 3: x + y) against the library GEMM followed by fg_gate_residual_bf16, and mode 4 (GELU) against the library's
       GELU epilogue (torch._addmm_activation)
  fp8: fg_gemm_fp8_bf16 against torch._scaled_mm (row-wise scale_a) on the same e4m3 operands
"""
import argparse
import ctypes
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fairygen_amd import hip  # noqa: E402


def timeit(fns, rounds):
    times = {kk: [] for kk in fns}
    for _ in range(2):
        for f in fns.values():
            f()
    torch.cuda.synchronize()
    for _ in range(rounds):
        for kk, f in fns.items():
            evs = []
            for _ in range(3):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record(); f(); e.record()
                evs.append((s, e))
            torch.cuda.synchronize()
            times[kk] += [s.elapsed_time(e) for s, e in evs]
    return {kk: sorted(ts) for kk, ts in times.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=27280)
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--only", default="")
    ap.add_argument("--shapes", default="qkv,o,ffn.0,ffn.2")
    ap.add_argument("--stamp-lib-p", default="", help="a tools/build_gemm_variant.sh build with --stamp / -DFG_GEMM_STAMP")
    a = ap.parse_args()
    hip.load()
    dev = "cuda"
    g = torch.Generator(dev).manual_seed(0)
    rnd = lambda *s, sc=1.0: (torch.randn(s, generator=g, device=dev, dtype=torch.float32) * sc).to(torch.bfloat16)  # noqa: E731
    shapes = [("qkv", 3072, 9216), ("o", 3072, 3072), ("ffn.0", 3072, 14336), ("ffn.2", 14336, 3072)]
    shapes = [s for s in shapes if s[0] in a.shapes.split(",")]
    stamp_p = None
    if a.stamp_lib_p:
        stamp_p = ctypes.CDLL(os.path.abspath(a.stamp_lib_p))
        stamp_p.fg_gemm_epilogue_bf16.restype = ctypes.c_int
        stamp_p.fg_gemm_epilogue_bf16.argtypes = hip._SIGNATURES["fg_gemm_epilogue_bf16"]
        stamp_p.fg_gemm_stamp_read.restype = ctypes.c_int
        stamp_p.fg_gemm_stamp_read.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    first_rows = min(a.m, 220)
    for name, k, n in shapes:
        x, w, b = rnd(a.m, k), rnd(n, k, sc=0.02), rnd(n, sc=0.1)
        ref = F.linear(x, w, b)
        rows = torch.randint(0, a.m, (64,), device=dev)
        ref32 = x[rows].float() @ w.float().t() + b.float()
        e_ref = (ref[rows].float() - ref32).abs().max().item()
        fns = {"hipBLASLt": lambda: F.linear(x, w, b)}
        if a.only in ("", "p"):
            outp = torch.full_like(ref, float("nan"))
            hip.gemm_epilogue(x, w, b, out=outp)
            torch.cuda.synchronize()
            e_p = (outp[rows].float() - ref32).abs().max().item()
            print(f"{name}: max|lib - f32| {e_ref:.4f}, max|p - f32| {e_p:.4f}, bit-identical to the library: "
                  f"{(outp == ref).float().mean().item():.6f}, NaNs left {torch.isnan(outp.float()).sum().item()}", flush=True)
            fns["p"] = lambda: hip.gemm_epilogue(x, w, b, out=outp)
            if n == 3072:      # residual modes (o-projections, ffn.2): against library GEMM + fg_gate_residual_bf16
                res0 = rnd(a.m, n)
                for rows_mod in (1, 2):
                    mod = hip.ModTable(rnd(rows_mod, 6, n), first_rows if rows_mod == 2 else 0)
                    want = hip.gate_residual(res0, ref, mod, 2)
                    got = hip.gemm_epilogue(x, w, b, out=res0.clone(), residual=True, mod=mod, gate_idx=2)
                    torch.cuda.synchronize()
                    print(f"   mode 2, {rows_mod}-row gate: equal to GEMM + gate_residual: {torch.equal(got, want)} "
                          f"({(got == want).float().mean().item():.6f})", flush=True)
                want = hip.gate_residual(res0, ref)
                got = hip.gemm_epilogue(x, w, b, out=res0.clone(), residual=True)
                torch.cuda.synchronize()
                print(f"   mode 3: equal to GEMM + residual: {torch.equal(got, want)} ({(got == want).float().mean().item():.6f})", flush=True)
                mod = hip.ModTable(rnd(2, 6, n), first_rows)
                resid = res0.clone()
                y_buf = torch.empty_like(ref)
                fns["hipBLASLt + gate_residual"] = lambda: hip.gate_residual(resid, torch.addmm(b, x, w.t(), out=y_buf), mod, 2, out=resid)
                fns["p mode 2"] = lambda: hip.gemm_epilogue(x, w, b, out=resid, residual=True, mod=mod, gate_idx=2)
            if name == "ffn.0":      # GELU(tanh) in the store against the library's GELU epilogue
                got = hip.gemm_epilogue(x, w, b, act="gelu_tanh")
                want = torch._addmm_activation(b, x, w.t(), use_gelu=True)
                torch.cuda.synchronize()
                d = (got.float() - want.float()).abs()
                print(f"   mode 4 (gelu): max |own - library epilogue| {d.max().item():.4f}, identical {(got == want).float().mean().item():.4f}", flush=True)
                outg = torch.empty_like(ref)
                fns["hipBLASLt + gelu epilogue"] = lambda: torch._addmm_activation(b, x, w.t(), use_gelu=True)
                fns["p mode 4 (gelu)"] = lambda: hip.gemm_epilogue(x, w, b, out=outg, act="gelu_tanh")
        if a.only in ("", "fp8"):
            xq, sc = hip.fp8_quant_rows(x)
            w8 = w.to(torch.float8_e4m3fn)
            ones = torch.ones((1, n), device=dev)
            lib8 = torch._scaled_mm(xq, w8.T, scale_a=sc, scale_b=ones, bias=b, out_dtype=torch.bfloat16)
            own8 = hip.gemm_fp8(xq, sc, w8, b)
            torch.cuda.synchronize()
            d = (own8.float() - lib8.float()).abs()
            print(f"{name} fp8: max |own - torch._scaled_mm| {d.max().item():.4f} (max |y| {lib8.float().abs().max().item():.2f}), identical "
                  f"{(own8 == lib8).float().mean().item():.5f}", flush=True)
            out8 = torch.empty_like(lib8)
            fns["fp8 torch._scaled_mm"] = lambda: torch._scaled_mm(xq, w8.T, scale_a=sc, scale_b=ones, bias=b, out_dtype=torch.bfloat16)
            fns["fp8 own"] = lambda: hip.gemm_fp8(xq, sc, w8, b, out=out8)
        times = timeit(fns, a.rounds)
        fl = 2.0 * a.m * k * n
        for kk, ts in times.items():
            print(f"   {kk}: median {ts[len(ts) // 2]:.3f} ms = {fl / ts[len(ts) // 2] / 1e9:.1f} TFLOP/s (min {ts[0]:.3f} ms)", flush=True)
        if stamp_p is not None:
            import numpy as np
            o2 = torch.zeros_like(ref)
            wsp = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
            st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            for _ in range(3):
                stamp_p.fg_gemm_epilogue_bf16(x.data_ptr(), k, w.data_ptr(), b.data_ptr(), o2.data_ptr(), n, a.m, n, k, 0, None, 1, n, 0, wsp.data_ptr(), st)
            buf = np.zeros((256, 4, 8), dtype=np.uint32)      # one workgroup per CU (256 on MI355X)
            stamp_p.fg_gemm_stamp_read(buf.ctypes.data_as(ctypes.c_void_p), st)
            w0 = buf[:, 0, :].astype(np.float64)          # wave 0 of every workgroup: kloop, epilogue, wait, tiles, total, ticks
            tiles = np.maximum(w0[:, 3], 1)              # units this workgroup took from the cursors
            clk = np.median(w0[:, 4] / np.maximum(w0[:, 5], 1) * 100)
            print(f"   p stamp ({'equal' if torch.equal(o2, outp) else 'DIFFERENT'} output): tiles/CU {tiles.min():.0f}-{tiles.max():.0f}; per tile: "
                  f"k-loop {np.median(w0[:, 0] / tiles):.0f} ({np.median(w0[:, 0] / tiles) / (k // 64):.0f} per 64-k step), epilogue "
                  f"{np.median(w0[:, 1] / tiles):.0f}, wait {np.median(w0[:, 2] / tiles):.0f} cycles; kernel {np.median(w0[:, 4]):.0f} cycles "
                  f"(max {w0[:, 4].max():.0f}), clock {clk:.0f} MHz; sum of parts {np.median((w0[:, 0] + w0[:, 1] + w0[:, 2])):.0f}", flush=True)
            tot = np.sort(w0[:, 4])
            print(f"      per-CU kernel cycles: p10 {tot[25]:.0f} p50 {tot[128]:.0f} p90 {tot[230]:.0f} p99 {tot[253]:.0f} max {tot[-1]:.0f}; units per CU histogram "
                  f"{dict(zip(*np.unique(w0[:, 3].astype(int), return_counts=True)))}; ideal (sum of all CUs / 256) {w0[:, 4].sum() / 256:.0f}", flush=True)


if __name__ == "__main__":
    main()

"""Diagnostic: where fg_gemm_fp8_bf16 differs from torch._scaled_mm / an fp64 evaluation of the same e4m3 operands."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fairygen_amd import hip

def seeded(shape, seed, scale=1.0):
    g = torch.Generator("cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g, dtype=torch.float32) * scale).to(torch.bfloat16)

for M, K, N, big in [(700, 3072, 768, True), (700, 3072, 768, False), (4200, 1024, 4096, True), (600, 14336, 3072, True), (27280, 3072, 3072, False)]:
    x = seeded((1, M, K), 151, 2.0)
    if big:
        x[0, 5] *= 300.0
        x[0, M - 3] *= 1000.0
    w, b = seeded((N, K), 152, 0.05).cuda(), seeded((N,), 153, 0.3).cuda()
    xq, sc = hip.fp8_quant_rows(x.cuda())
    w8 = w.to(torch.float8_e4m3fn)
    y = hip.gemm_fp8(xq, sc, w8, b)
    lib = torch._scaled_mm(xq, w8.T, scale_a=sc, scale_b=torch.ones((1, N), device="cuda"), bias=b, out_dtype=torch.bfloat16)
    rows = torch.cat([torch.arange(0, min(M, 512)), torch.arange(max(M - 512, 0), M)]).unique().cuda()
    exact = (xq[rows].double() @ w8.double().T) * sc[rows].double() + b.double()
    e_own, e_lib = (y[rows].double() - exact).abs(), (lib[rows].double() - exact).abs()
    ulp = torch.maximum(exact.abs(), torch.tensor(1e-3, device="cuda", dtype=torch.float64)) * 2.0 ** -8      # half-ulp-ish yardstick
    print(f"M={M} K={K} N={N} big={big}: own!=lib {(y != lib).float().mean().item():.5f}; err/yard own max {(e_own / ulp).max().item():.2f} "
          f"lib max {(e_lib / ulp).max().item():.2f}; own worse than 1.01 yard: {(e_own > 1.01 * ulp).sum().item()}, lib: {(e_lib > 1.01 * ulp).sum().item()}", flush=True)
    bad = (e_own > 1.5 * ulp).nonzero()
    if len(bad):
        r, c = bad[:, 0], bad[:, 1]
        print("   bad rows (index into sample) hist top:", torch.bincount(r).topk(min(5, len(torch.bincount(r)))), " distinct cols", c.unique().numel())
        print("   col % 64 hist:", torch.bincount(c % 64, minlength=64).tolist())
        i = bad[0]
        print("   first:", rows[i[0]].item(), i[1].item(), "own", y[rows][i[0], i[1]].item(), "lib", lib[rows][i[0], i[1]].item(), "exact", exact[i[0], i[1]].item(), "scale", sc[rows][i[0]].item())

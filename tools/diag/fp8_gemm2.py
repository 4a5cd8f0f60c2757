import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fairygen_amd import hip
from oracle import wan_dit

def seeded(shape, seed, scale=1.0):
    g = torch.Generator("cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g, dtype=torch.float32) * scale).to(torch.bfloat16)

M, K, N = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (700, 3072, 768)))
x = seeded((1, M, K), 151, 2.0)
x[0, 5] *= 300.0
x[0, M - 3] *= 1000.0
w, b = seeded((N, K), 152, 0.05), seeded((N,), 153, 0.3)
xq, sc = hip.fp8_quant_rows(x.cuda())
w8 = w.cuda().to(torch.float8_e4m3fn)
y = hip.gemm_fp8(xq, sc, w8, b.cuda()).cpu()
rows = torch.cat([torch.arange(0, 160), torch.arange(M - 160, M)])
want = wan_dit.scaled_mm(xq[rows.cuda()].cpu(), w8.cpu().T, sc[rows.cuda()].cpu(), torch.ones((1, N)), b, torch.bfloat16)
acc64 = xq[rows.cuda()].cpu().double() @ w8.cpu().double().T
acc32 = xq[rows.cuda()].cpu().float() @ w8.cpu().float().T
exact = acc64 * sc[rows.cuda()].cpu().double() + b.double()

got = y[rows]
mag = ((xq.float().abs() @ w8.float().abs().T) * sc * 2.0 ** -12).cpu()[rows]
for i in range(len(rows)):
    d = (got[i].float() - want[i].float()).abs()
    big = d > torch.maximum(want[i].float().abs(), mag[i]).clamp_min(1e-3) * 2.0 ** -7
    if big.any():
        print(f"row {rows[i].item()}: own != oracle on {(got[i] != want[i]).sum().item()} of {N}; beyond 1 ulp: {big.sum().item()}; scale {sc[rows[i]].item()}")
    for j in big.nonzero().flatten()[:6].tolist():
        print(f"    col {j}: own {got[i, j].item()} oracle {want[i, j].item()} exact {exact[i, j].item():.4f} acc64 {acc64[i, j].item():.4f} acc32(cpu) {acc32[i, j].item():.4f}")
# which of the two is closer to exact overall
e_own, e_or = (got.double() - exact).abs(), (want.double() - exact).abs()
print("mean |err| own", e_own.mean().item(), "oracle", e_or.mean().item())

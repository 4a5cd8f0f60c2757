"""hipBLASLt GEMM shapes of one DiT block at N tokens: time, TFLOP/s, and the GELU-epilogue variant."""
import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fairygen_amd import hip
from tools.microbench import timeit

n = int(sys.argv[1]) if len(sys.argv) > 1 else 27280
dev = "cuda"
g = torch.Generator(dev).manual_seed(0)
rnd = lambda *s: (torch.randn(s, generator=g, device=dev) * 0.05).to(torch.bfloat16)
x = rnd(1, n, 3072)
for name, k, m in [("qkv", 3072, 9216), ("o/q", 3072, 3072), ("ffn0", 3072, 14336), ("ffn2", 14336, 3072)]:
    a = rnd(1, n, k); w = rnd(m, k); b = rnd(m)
    med, _ = timeit(lambda: F.linear(a, w, b), 10)
    print(f"{name}: {med:.3f} ms  {2.0 * n * k * m / med / 1e9:.0f} TFLOP/s")
a = rnd(1, n, 3072); w = rnd(14336, 3072); b = rnd(14336)
ref = hip.activation(F.linear(a, w, b), "gelu_tanh")
med0, _ = timeit(lambda: hip.activation(F.linear(a, w, b), "gelu_tanh"), 10)
try:
    fused = torch._addmm_activation(b, a[0], w.t(), use_gelu=True)
    med1, _ = timeit(lambda: torch._addmm_activation(b, a[0], w.t(), use_gelu=True), 10)
    d = (fused.float() - ref[0].float()).abs()
    print(f"ffn0+gelu: linear+act {med0:.3f} ms; _addmm_activation {med1:.3f} ms; max diff {d.max().item():.4f} mean {d.mean().item():.6f}; mismatch frac {(fused != ref[0]).float().mean().item():.4f}")
except Exception as e:
    print("addmm_activation failed:", e)

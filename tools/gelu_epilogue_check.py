"""Which GELU does the hipBLASLt epilogue behind torch._addmm_activation(use_gelu=True) implement?  Compare its bf16
output with fp32 references of tanh-GELU (what nn.GELU(approximate='tanh') in the reference FFN computes) and erf-GELU."""
import torch, torch.nn.functional as F
torch.manual_seed(0)
n, k, m = 4096, 3072, 14336
a = (torch.randn(n, k, device="cuda") * 1.0).to(torch.bfloat16)
w = (torch.randn(m, k, device="cuda") * 0.03).to(torch.bfloat16)      # pre-activations ~ N(0, 1.7): both tails of GELU
b = (torch.randn(m, device="cuda") * 0.1).to(torch.bfloat16)
fused = torch._addmm_activation(b, a, w.t(), use_gelu=True).float()
pre = a.float() @ w.float().t() + b.float()
for name, ref in (("tanh", F.gelu(pre, approximate="tanh")), ("erf", F.gelu(pre))):
    d = (fused - ref).abs()
    print(f"{name}: mean abs diff {d.mean().item():.3e}  max {d.max().item():.3e}")
sel = (pre.abs() > 1.5) & (pre.abs() < 3.0)      # where tanh and erf GELU differ most (up to ~4.7e-4)
for name, ref in (("tanh", F.gelu(pre, approximate="tanh")), ("erf", F.gelu(pre))):
    print(f"{name} on 1.5<|x|<3: mean signed diff {(fused - ref)[sel].mean().item():.3e}")

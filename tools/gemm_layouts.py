"""Does the weight layout change the hipBLASLt GEMM rate?  F.linear (weight (N,K), K-contiguous: 'TN') vs x @ W with W stored
(K,N) N-contiguous ('NN'), bias added by addmm in both."""
import os, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.microbench import timeit
dev = "cuda"
g = torch.Generator(dev).manual_seed(0)
rnd = lambda *s: (torch.randn(s, generator=g, device=dev) * 0.05).to(torch.bfloat16)
m = int(sys.argv[1]) if len(sys.argv) > 1 else 27280
for name, k, n in [("qkv", 3072, 9216), ("o/q", 3072, 3072), ("ffn0", 3072, 14336), ("ffn2", 14336, 3072)]:
    x, w, b = rnd(m, k), rnd(n, k), rnd(n)
    wt = w.t().contiguous()
    t_tn = timeit(lambda: F.linear(x, w, b), 20)[0]
    t_nn = timeit(lambda: torch.addmm(b, x, wt), 20)[0]
    fl = 2.0 * m * k * n
    print(f"M={m} {name}: TN {t_tn:.3f} ms ({fl / t_tn / 1e9:.0f} TF/s)   NN {t_nn:.3f} ms ({fl / t_nn / 1e9:.0f} TF/s)")

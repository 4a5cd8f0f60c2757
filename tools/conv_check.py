"""Correctness of fg_conv3d_cl_bf16 (whatever form FAIRYGEN_CONV_TILE selects) against an fp32 torch convolution on the device."""
import os, sys, itertools
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fairygen_amd import hip
hip.load()
g = torch.Generator("cuda").manual_seed(0)
rnd = lambda *s, sc=1.0: (torch.randn(s, generator=g, device="cuda") * sc).to(torch.bfloat16)
cases = [(128, 256, 3, 1, 1, 5, 6, False), (64, 256, 1, 1, 1, 5, 6, False), (64, 256, 1, 1, 2, 16, 16, False), (128, 256, 1, 1, 2, 16, 16, False),
         (64, 256, 3, 1, 2, 16, 16, False), (64, 256, 1, 3, 2, 16, 16, False), (256, 256, 3, 3, 2, 16, 16, True), (1024, 1024, 3, 3, 1, 4, 6, False),
         (256, 512, 3, 3, 4, 30, 52, True)]
for cin, cout, kt, ks, T, H, W, res in cases:
    x = rnd(T + kt - 1, H, W, cin)
    w = rnd(cout, cin, kt, ks, ks, sc=(cin * kt * ks * ks) ** -0.5)
    b = rnd(cout, sc=0.1)
    r = rnd(T, H, W, cout) if res else None
    got = hip.conv3d_cl(x, hip.conv_pack_weight(w), b, cout, kt, ks, residual=r).float()
    xin = x.permute(3, 0, 1, 2).unsqueeze(0).float()
    ref = F.conv3d(F.pad(xin, (ks // 2, ks // 2, ks // 2, ks // 2, 0, 0)), w.float(), b.float())[0].permute(1, 2, 3, 0)
    if res:
        ref = ref + r.float()
    err = (got - ref).abs()
    bad = (err > 0.06).nonzero()
    print(f"cin {cin} cout {cout} kt {kt} ks {ks} T {T} H {H} W {W} res {res}: max err {err.max().item():.4f}, bad {len(bad)} / {err.numel()}",
          (bad[:3].tolist() if len(bad) else ""), flush=True)

"""Diagnostic: the w4 attention kernel on short KV ranges (cross-attention: 512 keys) — correctness against an fp32 evaluation and time
against attn_fwd_kernel<8,1,true>.  Run twice: default, and FAIRYGEN_ATTN_W4_MIN_KV=64."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fairygen_amd import hip

g = torch.Generator("cuda").manual_seed(0)
rnd = lambda *s: torch.randn(s, generator=g, device="cuda", dtype=torch.float32).to(torch.bfloat16)
for nkv in (64, 100, 128, 192, 256, 333, 512, 1000):
    heads, nq = 2, 700
    c = heads * 128
    q, k, v = rnd(1, nq, c), rnd(1, nkv, c), rnd(1, nkv, c)
    got = hip.attention(q, k, v, heads)[0].float()
    qh, kh, vh = (t[0].view(-1, heads, 128).float().transpose(0, 1) for t in (q, k, v))
    ref = (torch.softmax(qh @ kh.transpose(1, 2) * 128 ** -0.5, -1) @ vh).transpose(0, 1).reshape(nq, c)
    print(f"nkv={nkv}: max err {(got - ref).abs().max().item():.5f} finite {bool(torch.isfinite(got).all())}", flush=True)
heads, nq, nkv = 24, 27280, 512
c = heads * 128
q, k, v = rnd(1, nq, c), rnd(1, nkv, c), rnd(1, nkv, c)
out = torch.empty_like(q)
for _ in range(3):
    hip.attention(q, k, v, heads, out=out)
ts = []
for _ in range(10):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); hip.attention(q, k, v, heads, out=out); e.record(); torch.cuda.synchronize()
    ts.append(s.elapsed_time(e))
t = sorted(ts)[5]
print(f"cross-attention shape nq={nq} nkv={nkv}: {t:.3f} ms = {4.0 * nq * nkv * c / t / 1e9:.0f} TFLOP/s", flush=True)

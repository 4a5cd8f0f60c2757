"""Probe: does torch._scaled_mm take the reference's fp8_linear call (row-wise scale_a, unit scale_b, bf16 bias) on gfx950,
with which fp8 flavour, and how fast is it next to the bf16 GEMM?"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.microbench import timeit
dev = "cuda"
g = torch.Generator(dev).manual_seed(0)
m, k, n = 27280, 3072, 14336
x = (torch.randn(m, k, generator=g, device=dev)).to(torch.bfloat16)
w = (torch.randn(n, k, generator=g, device=dev) * 0.02).to(torch.bfloat16)
b = (torch.randn(n, generator=g, device=dev) * 0.1).to(torch.bfloat16)
ref = torch.nn.functional.linear(x, w, b)
for dt in (torch.float8_e4m3fn, torch.float8_e4m3fnuz):
    try:
        fmax = 448.0 if dt == torch.float8_e4m3fn else 224.0
        x_max = x.abs().amax(dim=-1, keepdim=True)
        scale_a = torch.clamp(x_max / fmax, min=1.0).float()
        scale_b = torch.ones((n, 1), device=dev)
        xq = (x / (scale_a + 1e-8)).to(dt)
        wq = w.to(dt)
        out = torch._scaled_mm(xq, wq.T, scale_a=scale_a, scale_b=scale_b.T, bias=b, out_dtype=torch.bfloat16)
        err = (out.float() - ref.float()).abs().mean().item() / ref.float().abs().mean().item()
        med, _ = timeit(lambda: torch._scaled_mm(xq, wq.T, scale_a=scale_a, scale_b=scale_b.T, bias=b, out_dtype=torch.bfloat16), 10)
        print(f"{dt}: rowwise ok, rel mean err vs bf16 {err:.4f}, {med:.3f} ms = {2.0 * m * k * n / med / 1e9:.0f} TFLOP/s")
        # tensor-wise variant
        s1 = torch.ones((), device=dev)
        out2 = torch._scaled_mm(xq, wq.T, scale_a=s1, scale_b=s1, bias=b, out_dtype=torch.bfloat16)
        med2, _ = timeit(lambda: torch._scaled_mm(xq, wq.T, scale_a=s1, scale_b=s1, bias=b, out_dtype=torch.bfloat16), 10)
        print(f"{dt}: tensorwise {med2:.3f} ms = {2.0 * m * k * n / med2 / 1e9:.0f} TFLOP/s; same as rowwise (scale 1): {torch.equal(out, out2)}")
    except Exception as e:
        print(f"{dt}: FAILED {type(e).__name__}: {str(e)[:300]}")
med, _ = timeit(lambda: torch.nn.functional.linear(x, w, b), 10)
print(f"bf16: {med:.3f} ms = {2.0 * m * k * n / med / 1e9:.0f} TFLOP/s")

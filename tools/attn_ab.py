"""A/B of fg_attn_fwd_bf16 builds in ONE process (interleaved rounds, random data, HIP events on the launch stream).

    python tools/attn_ab.py --libs base=fairygen_amd/libfairygen_hip.so,new=fairygen_amd/csrc/build/ab/libfg_new.so \
        [--nq 27280 --nkv 27280 --heads 24 --rounds 8 --check-rows 192]

Every library is checked first against an fp32 reference (sampled query rows x all keys, every head) with the criterion of
tests/test_hip_kernels.py (max|out - ref_f32| against the bf16 rounding of the reference), then timed round-robin.
"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fairygen_amd import hip  # noqa: E402

_i64, _i32, _f32, _vp = ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_void_p


def load(path):
    lib = ctypes.CDLL(os.path.abspath(path))
    lib.fg_attn_fwd_bf16.restype = ctypes.c_int
    lib.fg_attn_fwd_bf16.argtypes = hip._SIGNATURES["fg_attn_fwd_bf16"]
    lib.fg_attn_workspace_bytes.restype = _i64
    lib.fg_attn_workspace_bytes.argtypes = [_i32, _i64, _i64, _i32]
    lib.fg_last_error.restype = ctypes.c_char_p
    return lib


def runner(lib, q, k, v, heads, out):
    b, nq, hd = q.shape
    nkv = k.shape[1]
    need = lib.fg_attn_workspace_bytes(b, nq, nkv, heads)
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device=q.device)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def fn():
        rc = lib.fg_attn_fwd_bf16(q.data_ptr(), q.stride(1), k.data_ptr(), k.stride(1), v.data_ptr(), v.stride(1), out.data_ptr(),
                                  b, nq, nkv, heads, hd // heads, float(hd // heads) ** -0.5, ws.data_ptr() if need else None, need, stream)
        if rc:
            raise RuntimeError(lib.fg_last_error().decode())
    return fn


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--libs", default="base=" + hip.library_path())
    ap.add_argument("--nq", type=int, default=27280)
    ap.add_argument("--nkv", type=int, default=27280)
    ap.add_argument("--heads", type=int, default=24)
    ap.add_argument("--rounds", type=int, default=8)
    ap.add_argument("--iters", type=int, default=3, help="launches per library per round")
    ap.add_argument("--check-rows", type=int, default=192)
    ap.add_argument("--stamp", default="", help="comma list of libraries built with gen_attn_w4.py --stamp: decode their cycle stamps")
    ap.add_argument("--spike", action="store_true", help="also check an input that forces the deferred-rescale branch")
    a = ap.parse_args()
    dev = "cuda"
    g = torch.Generator(dev).manual_seed(0)
    c = a.heads * 128
    rnd = lambda *s: torch.randn(s, generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)  # noqa: E731
    q, k, v = rnd(1, a.nq, c), rnd(1, a.nkv, c), rnd(1, a.nkv, c)
    libs = [(n, load(p)) for n, p in (item.split("=", 1) for item in a.libs.split(","))]
    rows = torch.randperm(a.nq, generator=g, device=dev)[: a.check_rows].sort().values
    rows = torch.cat([rows, torch.tensor([0, a.nq - 1], device=dev)]).unique()

    def reference(q, k, v):
        qs = q[0, rows].view(-1, a.heads, 128).float().transpose(0, 1)             # (H, R, 128)
        kk = k[0].view(-1, a.heads, 128).float().permute(1, 2, 0)                  # (H, 128, N)
        p = torch.softmax(torch.bmm(qs, kk) * 128 ** -0.5, dim=-1)
        return torch.bmm(p, v[0].view(-1, a.heads, 128).float().transpose(0, 1)).transpose(0, 1).reshape(-1, c)

    cases = [("random", q, k, v)]
    if a.spike:      # one key row aligned with a few query rows at growing magnitude: the running max jumps past the defer threshold
        k2 = k.clone()
        for i, pos in enumerate((a.nkv // 3, a.nkv // 2, a.nkv - 5)):
            k2[0, pos] = (q[0, rows[i % len(rows)]].float() * (1.5 + i)).to(torch.bfloat16)
        cases.append(("spiked", q, k2, v))
    for cname, cq, ck, cv in cases:
        ref = reference(cq, ck, cv)
        ref_bf = ref.to(torch.bfloat16).float()
        err_ref = (ref_bf - ref).abs().max().item()
        for name, lib in libs:
            out = torch.zeros((1, a.nq, c), dtype=torch.bfloat16, device=dev)
            runner(lib, cq, ck, cv, a.heads, out)()
            torch.cuda.synchronize()
            got = out[0, rows].float()
            err = (got - ref).abs().max().item()
            mean = (got - ref).abs().mean().item()
            ok = err <= 2 * err_ref + 2e-2 and bool(torch.isfinite(out.float()).all())
            print(f"check[{cname}] {name}: max|out - ref_f32| = {err:.5f} (bf16 rounding of ref: {err_ref:.5f}), mean {mean:.6f}  "
                  f"{'OK' if ok else 'FAIL'}", flush=True)
    for name, lib in libs:
        if name not in a.stamp.split(","):
            continue
        out = torch.zeros((1, a.nq, c), dtype=torch.bfloat16, device=dev)
        fn = runner(lib, q, k, v, a.heads, out)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        raw = out[0, 0::256, :].contiguous().view(torch.int32).view(-1, a.heads, 64)[:, :, :4].reshape(-1, 4).cpu()
        raw = raw[raw[:, 2] > 0]
        cyc, ticks, steps = raw[:, 0].double(), raw[:, 1].double(), raw[:, 2].double()
        per_step = (cyc / steps).median().item()
        clock = (cyc / ticks * 100).median().item()
        outside = (raw[:, 3].double() - cyc).median().item()
        print(f"stamp {name}: {len(raw)} workgroups, median {per_step:.0f} shader cycles per step (64 MFMAs = 2048), "
              f"in-kernel clock {clock:.0f} MHz, steps {int(steps.median().item())}, prologue+epilogue {outside:.0f} cycles "
              f"(loop total {cyc.median().item():.0f})", flush=True)
    out = torch.empty((1, a.nq, c), dtype=torch.bfloat16, device=dev)
    fns = [(n, runner(lib, q, k, v, a.heads, out)) for n, lib in libs]
    for _, fn in fns:
        for _ in range(2):
            fn()
    torch.cuda.synchronize()
    times = {n: [] for n, _ in fns}
    for _ in range(a.rounds):
        for n, fn in fns:
            evs = []
            for _ in range(a.iters):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record(); fn(); e.record()
                evs.append((s, e))
            torch.cuda.synchronize()
            times[n] += [s.elapsed_time(e) for s, e in evs]
    fl = 4.0 * a.nq * a.nkv * c
    for n, ts in times.items():
        ts = sorted(ts)
        med, mn = ts[len(ts) // 2], ts[0]
        print(f"{n}: nq={a.nq} nkv={a.nkv} H={a.heads}: median {med:.3f} ms = {fl / med / 1e9:.1f} TFLOP/s, min {mn:.3f} ms = {fl / mn / 1e9:.1f}", flush=True)


if __name__ == "__main__":
    main()

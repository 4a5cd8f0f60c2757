"""What the persistent GEMM costs when some CUs are not available to it (another stream's kernel — RCCL — holds them): the same launch
with fewer workgroups than CUs (fg_gemm_debug_grid).  Units are taken from per-XCD cursors, so the time should grow like CUs / workgroups,
not jump to the next whole round as a statically dealt tile list would.

    python tools/gemm_busy_cus.py [--m 27280] [--grids 256,248,240,224,192,128]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fairygen_amd import hip  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=27280)
    ap.add_argument("--grids", default="256,248,240,224,192,128")
    a = ap.parse_args()
    lib = hip.load()
    g = torch.Generator("cuda").manual_seed(0)
    rnd = lambda *s, sc=1.0: (torch.randn(s, generator=g, device="cuda", dtype=torch.float32) * sc).to(torch.bfloat16)  # noqa: E731
    for name, k, n in (("qkv", 3072, 9216), ("o", 3072, 3072), ("ffn.2", 14336, 3072)):
        x, w, b = rnd(a.m, k), rnd(n, k, sc=0.02), rnd(n, sc=0.1)
        out = torch.empty((a.m, n), dtype=torch.bfloat16, device="cuda")
        want = hip.gemm_epilogue(x, w, b).clone()
        base = None
        for grid in [int(v) for v in a.grids.split(",")]:
            lib.fg_gemm_debug_grid(grid)
            for _ in range(2):
                hip.gemm_epilogue(x, w, b, out=out)
            ts = []
            for _ in range(8):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record(); hip.gemm_epilogue(x, w, b, out=out); e.record()
                torch.cuda.synchronize()
                ts.append(s.elapsed_time(e))
            t = sorted(ts)[len(ts) // 2]
            base = base or t
            print(f"{name} M={a.m}: {grid:3d} workgroups: {t:.3f} ms = x{t / base:.3f} of the full grid (CUs / workgroups = {256 / grid:.3f}); "
                  f"result {'identical' if torch.equal(out, want) else 'DIFFERENT'}", flush=True)
        lib.fg_gemm_debug_grid(0)


if __name__ == "__main__":
    main()

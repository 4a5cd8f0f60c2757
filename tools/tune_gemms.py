"""hipBLASLt solution search (PyTorch TunableOp) for the DiT block's GEMM shapes at the token counts the multi-GPU
layouts produce (N, N/2, N/4, N/8 tokens per rank).  Keeps only shapes where a non-default solution is clearly faster
(--gain, default 7 %) and writes them as a TunableOp results file: fairygen_amd/tuning/gfx950_gemm.csv, which
fairygen_amd.tuning loads read-only at run time.

    python tools/tune_gemms.py [--tokens 27280,5070] [--out gpurun_out/gfx950_gemm.csv]
"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.microbench import timeit  # noqa: E402

SHAPES = [("qkv", 3072, 9216), ("o/q", 3072, 3072), ("ffn0", 3072, 14336), ("ffn2", 14336, 3072)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tokens", default="27280,5070")
    ap.add_argument("--out", default="gpurun_out/gfx950_gemm.csv")
    ap.add_argument("--gain", type=float, default=0.07)
    a = ap.parse_args()
    dev = "cuda"
    g = torch.Generator(dev).manual_seed(0)
    rnd = lambda *s: (torch.randn(s, generator=g, device=dev) * 0.05).to(torch.bfloat16)  # noqa: E731
    ms = sorted({(int(t) + p - 1) // p for t in a.tokens.split(",") for p in (1, 2, 4, 8)}, reverse=True)
    base = {}
    for m in ms:
        for name, k, n in SHAPES:
            x, w, b = rnd(1, m, k), rnd(n, k), rnd(n)
            base[f"tn_{n}_{m}_{k}_ld_{k}_{k}_{n}"] = (timeit(lambda: F.linear(x, w, b), 20)[0], m, name)
    import torch.cuda.tunable as tunable
    tunable.enable(True)
    tunable.tuning_enable(True)
    tunable.set_filename(os.path.join(os.environ.get("TMPDIR", "/tmp"), "tune_gemms_scratch.csv"))
    tunable.set_max_tuning_iterations(50)
    for m in ms:
        for name, k, n in SHAPES:
            x, w, b = rnd(1, m, k), rnd(n, k), rnd(n)
            F.linear(x, w, b)
    torch.cuda.synchronize()
    kept = []
    for op, params, solution, t in tunable.get_results():
        t0, m, name = base.get(params, (None, None, None))
        if t0 is None:
            continue
        keep = solution != "Default" and t < (1.0 - a.gain) * t0
        print(f"M={m:6d} {name:5s}: default {t0:.3f} ms -> {solution} {t:.3f} ms {'KEEP' if keep else ''}", flush=True)
        if keep:
            kept.append(f"{op},{params},{solution},{t:.6g}")
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    with open(a.out, "w") as f:
        for v in tunable.get_validators():
            f.write("Validator," + ",".join(str(s) for s in v) + "\n")
        f.write("\n".join(kept) + "\n")
    print(open(a.out).read())


if __name__ == "__main__":
    main()

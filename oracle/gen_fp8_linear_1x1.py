"""Golden vectors for the fp8 Linear mode from the REFERENCE's own code (build container only; test infrastructure).

    python oracle/gen_fp8_linear_1x1.py          # needs /root/reference; writes tests/golden/fp8_linear_1x1.safetensors

AutoWrappedLinear.fp8_linear (animation/diffsynth/core/vram/layers.py:321-357) calls torch._scaled_mm with a (rows, 1) activation
scale and a (1, out) weight scale; this container's CPU backend accepts that only when both have ONE element (VERDICT r1 item 5:
the degenerate 1-row x 1-output call).  That call still runs every line of the method — per-row dynamic scale, division by
(scale + 1e-8), cast of activation and weight to e4m3fn, the scaled product, bf16 bias, rounding to the input dtype — so a batch of
such calls (different rows, weights, biases; reductions of the two lengths the DiT blocks use; rows above and below the fp8 range)
pins the arithmetic the fp8 mode is built from.  Only tensors are stored; the method is called unmodified, unbound, with an object
that carries the one attribute it reads (computation_dtype).
"""
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_golden as gg  # noqa: E402


def main():
    gg.import_reference()
    from diffsynth.core.vram.layers import AutoWrappedLinear
    self_like = types.SimpleNamespace(computation_dtype=torch.float8_e4m3fn)
    tensors, n = {}, 0
    for k, cases in ((3072, 48), (14336, 16)):
        xs, ws, bs, outs = [], [], [], []
        for i in range(cases):
            x = gg.seeded((1, k), 9000 + 10 * n)
            if i % 3 == 1:
                x[0, (7 * i) % k] = 300.0 + 40.0 * i             # row maximum around / above fp8_max = 448: scale_a >= 1
            if i % 3 == 2:
                x = (x.float() * 200.0).to(torch.bfloat16)       # the whole row far above the range: scale_a >> 1
            w = gg.seeded((1, k), 9001 + 10 * n, scale=0.05)
            b = gg.seeded((1,), 9002 + 10 * n, scale=0.5)
            with torch.no_grad():
                out = AutoWrappedLinear.fp8_linear(self_like, x, w, b)
            assert out.shape == (1, 1) and out.dtype == torch.bfloat16
            xs.append(x), ws.append(w), bs.append(b), outs.append(out.reshape(1))
            n += 1
        tensors[f"x_{k}"], tensors[f"w_{k}"] = torch.cat(xs), torch.cat(ws)
        tensors[f"b_{k}"], tensors[f"out_{k}"] = torch.cat(bs), torch.cat(outs)
    gg.save("fp8_linear_1x1.safetensors", tensors,
            {"source": "diffsynth/core/vram/layers.py AutoWrappedLinear.fp8_linear, one (1,K) row x one (1,K) weight row per case",
             "torch": torch.__version__})


if __name__ == "__main__":
    main()

"""The reference's tiled VAE38 decode with inference.py's tile configuration ((30,52)/(15,26)) at full decoder width on a latent
large enough for a 2 x 2 tile grid with feathered overlaps: (1,48,2,40,60) -> (1,3,5,640,960).  Reference vs oracle on the CPU.

    python oracle/gen_vae_tiled_full.py        # build container only (needs /root/reference); several minutes on 8 cores

Writes tests/golden/vae_tiled_full.safetensors: the decoded video subsampled by 8 in both spatial axes (offset 3, so tile seams
and borders are hit), bf16 from the reference and fp32 from the restatement.
"""
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))
import gen_golden  # noqa: E402


def main():
    torch.set_num_threads(8)
    R = gen_golden.import_reference()
    from fairygen_amd import synthetic
    from oracle import wan_vae
    vsd = synthetic.random_state_dict(synthetic.vae_shapes(), seed=1234)
    vae = R["vae"].WanVideoVAE38()
    vae.load_state_dict(vsd, assign=True)
    vae = vae.to(torch.bfloat16).eval()
    z = gen_golden.seeded((1, 48, 2, 40, 60), 35)
    timing = {}
    with torch.no_grad():
        t0 = time.perf_counter()
        ref = vae.decode(z, device="cpu", tiled=True, tile_size=(30, 52), tile_stride=(15, 26))
        timing["reference_s"] = time.perf_counter() - t0
        print(f"reference: {timing['reference_s']:.0f} s {tuple(ref.shape)}", flush=True)
        t0 = time.perf_counter()
        got = wan_vae.vae_decode(vsd, z, True, (30, 52), (15, 26))
        timing["oracle_s"] = time.perf_counter() - t0
        same = torch.equal(got, ref)
        print(f"oracle: {timing['oracle_s']:.0f} s equal {same}", flush=True)
        assert same
        del vae
        f32 = wan_vae.vae_decode({k: v.float() for k, v in vsd.items()}, z.float(), True, (30, 52), (15, 26))
    sub = lambda v: v[..., 3::8, 3::8].contiguous()  # noqa: E731
    timing.update(equal=same, bf16_vs_f32_max_abs=(ref.float() - f32).abs().max().item())
    print(json.dumps(timing), flush=True)
    gen_golden.save("vae_tiled_full.safetensors", {"video_bf16_sub8": sub(ref), "video_f32_sub8": sub(f32)}, {
        "config": "WanVideoVAE38() full width; decode(tiled=True, tile_size=(30,52), tile_stride=(15,26))",
        "weights": "synthetic.random_state_dict(vae_shapes(), seed=1234), CPU generator",
        "inputs": "z=seeded((1,48,2,40,60),35); stored video[..., 3::8, 3::8]", "timing": json.dumps(timing),
        "source": "diffsynth/models/wan_video_vae.py WanVideoVAE.decode / tiled_decode / build_mask :1081-1152,1235-1247"})


if __name__ == "__main__":
    main()

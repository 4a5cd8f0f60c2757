"""The VAE38 ENCODER at full width (dim 160: the real Wan2.2_VAE shapes; synthetic weights) with the REFERENCE's own
WanVideoVAE38.encode on the CPU next to the oracle: a 256x256 first-frame image (the TI2V conditioning path) and a 9-frame clip
(video-to-video: temporal stride-2 convolutions and caches).

    python oracle/gen_vae_encode_full.py        # build container only (needs /root/reference)
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
    img = gen_golden.seeded((3, 1, 256, 256), 32, scale=0.5).clamp(-1, 1)
    vid = gen_golden.seeded((3, 9, 128, 192), 33, scale=0.5).clamp(-1, 1)
    out, timing = {}, {}
    with torch.no_grad():
        for name, x in (("image", img), ("video", vid)):
            t0 = time.perf_counter()
            ref = vae.encode([x], device="cpu")
            timing[f"reference_{name}_s"] = time.perf_counter() - t0
            t0 = time.perf_counter()
            got = wan_vae.vae_encode(vsd, [x])
            timing[f"oracle_{name}_s"] = time.perf_counter() - t0
            assert torch.equal(got, ref), name
            out[f"encode_{name}_bf16"] = ref
            out[f"encode_{name}_f32"] = wan_vae.vae_encode({k: v.float() for k, v in vsd.items()}, [x.float()])
            print(name, tuple(ref.shape), json.dumps(timing), flush=True)
    gen_golden.save("vae_encode_full.safetensors", out, {
        "config": "WanVideoVAE38() full width", "weights": "synthetic.random_state_dict(vae_shapes(), seed=1234), CPU generator",
        "inputs": "img=seeded((3,1,256,256),32,scale=0.5).clamp(-1,1); vid=seeded((3,9,128,192),33,scale=0.5).clamp(-1,1)",
        "timing": json.dumps(timing), "source": "diffsynth/models/wan_video_vae.py WanVideoVAE.encode :1218-1232, VideoVAE38_.encode :1298-1323"})


if __name__ == "__main__":
    main()

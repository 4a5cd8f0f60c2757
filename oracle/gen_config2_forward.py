"""One DiT forward of BASELINE.json configs[1] (480x832x49: latent (1,48,13,30,52), N = 5 070 tokens — a ragged 19.8 query
blocks) with the REFERENCE's own model_fn_wan_video at full width on the CPU, next to the oracle in bf16 and fp32.

    python oracle/gen_config2_forward.py        # build container only (needs /root/reference); a few minutes on 8 cores

Writes tests/golden/config2_forward.safetensors: every 6th channel of the reference's bf16 prediction and of the fp32
evaluation (the yardstick of the GPU test's 2x criterion).
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
    from fairygen_amd.loader import TI2V_5B_DIT_KWARGS
    from oracle import wan_dit
    cfg = dict(TI2V_5B_DIT_KWARGS)
    sd = synthetic.random_state_dict(synthetic.dit_shapes(cfg), seed=1234)
    with torch.device("meta"):
        model = R["dit"].WanModel(**cfg)
    model.load_state_dict(sd, assign=True)
    model.freqs = R["dit"].precompute_freqs_cis_3d(cfg["dim"] // cfg["num_heads"])
    model.eval()
    s = gen_golden.seeded
    lat = s((1, 48, 13, 30, 52), 1)
    lat[:, :, 0:1] = s((1, 48, 1, 30, 52), 4)
    ctx = s((1, 512, 4096), 2); ctx[:, 64:] = 0
    ts = torch.tensor([700.0]).to(torch.bfloat16)
    timing = {}
    with torch.no_grad():
        t0 = time.perf_counter()
        ref = R["pipe"].model_fn_wan_video(dit=model, latents=lat, timestep=ts, context=ctx, fuse_vae_embedding_in_latents=True)
        timing["reference_forward_s"] = time.perf_counter() - t0
        print(f"reference forward: {timing['reference_forward_s']:.0f} s", flush=True)
        t0 = time.perf_counter()
        got = wan_dit.model_fn(sd, cfg, lat, ts, ctx, True)
        timing["oracle_forward_s"] = time.perf_counter() - t0
        same = torch.equal(got, ref)
        print(f"oracle forward: {timing['oracle_forward_s']:.0f} s, equals reference: {same}", flush=True)
        assert same, "the oracle restatement differs from the reference at N = 5070"
        del model
        f32 = wan_dit.model_fn({k: v.float() for k, v in sd.items()}, cfg, lat.float(), ts.float(), ctx.float(), True)
    timing.update(oracle_equals_reference=same, cores=torch.get_num_threads(),
                  bf16_vs_f32_max_abs=(ref.float() - f32).abs().max().item())
    print(json.dumps(timing, indent=1), flush=True)
    gen_golden.save("config2_forward.safetensors", {"pred_bf16_ch6": ref[:, ::6].contiguous(), "pred_f32_ch6": f32[:, ::6].contiguous()}, {
        "config": "TI2V_5B_DIT_KWARGS (30 blocks, dim 3072)", "weights": "synthetic.random_state_dict(dit_shapes(), seed=1234), CPU generator",
        "inputs": "latents=seeded((1,48,13,30,52),1) with frame 0 = seeded((1,48,1,30,52),4); ctx=seeded((1,512,4096),2) rows>=64 zero; "
                  "timestep=bf16(700); fuse_vae_embedding_in_latents=True; stored: channels 0,6,...,42",
        "timing": json.dumps(timing), "source": "diffsynth/pipelines/wan_video.py model_fn_wan_video :1122-1388"})


if __name__ == "__main__":
    main()

"""One DiT forward of BASELINE.json configs[2] — the headline configuration: 704x1280x121, latent (1,48,31,44,80),
N = 27 280 tokens, rank-32 merged motion LoRA fused at load — with the REFERENCE's own model_fn_wan_video and the REFERENCE's
own GeneralLoRALoader.fuse_lora_to_base_model at full width on the CPU, next to the oracle in bf16 and fp32.

    python oracle/gen_config3_forward.py        # build container only (needs /root/reference); ~25 min on 8 cores, ~45 GB

Writes tests/golden/config3_forward.safetensors: channels 0,8,...,40 x every 2nd latent column of the reference's bf16
prediction and of the fp32 evaluation of the same fused weights (the yardstick of the GPU test's 2x criterion).
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

LAT = (1, 48, 31, 44, 80)


def inputs():
    s = gen_golden.seeded
    lat = s(LAT, 1)
    lat[:, :, 0:1] = s((1, 48, 1, LAT[3], LAT[4]), 4)
    ctx = s((1, 512, 4096), 2)
    ctx[:, 64:] = 0
    ts = torch.tensor([700.0]).to(torch.bfloat16)
    return lat, ctx, ts


def main():
    torch.set_num_threads(8)
    R = gen_golden.import_reference()
    from fairygen_amd import synthetic
    from fairygen_amd.loader import TI2V_5B_DIT_KWARGS
    from oracle import wan_dit
    cfg = dict(TI2V_5B_DIT_KWARGS)
    shapes = synthetic.dit_shapes(cfg)
    sd = synthetic.random_state_dict(shapes, seed=1234)
    lora = synthetic.random_lora(shapes, rank=32, seed=4321)
    before = {k: v.float().sum().item() for k, v in sd.items()}       # load_state_dict(assign=True) shares storage with sd
    with torch.device("meta"):
        model = R["dit"].WanModel(**cfg)
    model.load_state_dict(sd, assign=True)
    model.freqs = R["dit"].precompute_freqs_cis_3d(cfg["dim"] // cfg["num_heads"])
    model.eval()
    loader = R["lora"](device="cpu", torch_dtype=torch.bfloat16)
    loader.fuse_lora_to_base_model(model, loader.convert_state_dict(lora), alpha=1)          # utils/lora/general.py:44-62
    fused = {k: v.detach() for k, v in model.state_dict().items() if k in shapes}
    changed = sum(int(fused[k].float().sum().item() != before[k]) for k in fused)
    print(f"reference LoRA fuse changed {changed} tensors", flush=True)
    assert changed == 300
    del sd
    lat, ctx, ts = inputs()
    timing = {}
    with torch.no_grad():
        t0 = time.perf_counter()
        ref = R["pipe"].model_fn_wan_video(dit=model, latents=lat, timestep=ts, context=ctx, fuse_vae_embedding_in_latents=True)
        timing["reference_forward_s"] = time.perf_counter() - t0
        print(f"reference forward: {timing['reference_forward_s']:.0f} s", flush=True)
        t0 = time.perf_counter()
        got = wan_dit.model_fn(fused, cfg, lat, ts, ctx, True)
        timing["oracle_forward_s"] = time.perf_counter() - t0
        same = torch.equal(got, ref)
        print(f"oracle forward: {timing['oracle_forward_s']:.0f} s, equals reference: {same}", flush=True)
        assert same, "the oracle restatement differs from the reference at N = 27280"
        del model, got
        f32sd = {k: v.float() for k, v in fused.items()}
        del fused
        t0 = time.perf_counter()
        f32 = wan_dit.model_fn(f32sd, cfg, lat.float(), ts.float(), ctx.float(), True)
        timing["oracle_f32_forward_s"] = time.perf_counter() - t0
    timing.update(oracle_equals_reference=same, cores=torch.get_num_threads(),
                  bf16_vs_f32_max_abs=(ref.float() - f32).abs().max().item(),
                  bf16_vs_f32_mean_abs=(ref.float() - f32).abs().mean().item())
    print(json.dumps(timing, indent=1), flush=True)
    gen_golden.save("config3_forward.safetensors",
                    {"pred_bf16_sub": ref[:, ::8, :, :, ::2].contiguous(), "pred_f32_sub": f32[:, ::8, :, :, ::2].contiguous()}, {
        "config": "TI2V_5B_DIT_KWARGS (30 blocks, dim 3072) + rank-32 LoRA fused by the reference's GeneralLoRALoader, alpha=1",
        "weights": "synthetic.random_state_dict(dit_shapes(), seed=1234) and synthetic.random_lora(shapes, rank=32, seed=4321), CPU generator",
        "inputs": "latents=seeded((1,48,31,44,80),1) with frame 0 = seeded((1,48,1,44,80),4); ctx=seeded((1,512,4096),2) rows>=64 zero; "
                  "timestep=bf16(700); fuse_vae_embedding_in_latents=True; stored: [:, ::8, :, :, ::2]",
        "timing": json.dumps(timing),
        "source": "diffsynth/pipelines/wan_video.py model_fn_wan_video :1122-1388; diffsynth/utils/lora/general.py:44-62"})


if __name__ == "__main__":
    main()

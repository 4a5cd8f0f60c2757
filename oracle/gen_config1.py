"""BASELINE.json configs[0] on the CPU with the REFERENCE's own code at full model width, next to the oracle restatement.

    python oracle/gen_config1.py            # build container only (needs /root/reference); ~10-20 min on 8 cores

Wan2.2-TI2V-5B (30 blocks, dim 3072: the synthetic random-init checkpoint with the reference's key/shape set) on a 256x256x17
clip = latent (1,48,5,16,16), 320 tokens, 4 denoise steps, CFG 5, shift 5, TI2V first-frame pin, then the VAE38 decode with
tiled=True.  Does three things (SURVEY.md §8d "CPU baseline beside it"):
  1. pins the oracle at FULL width: the restatement must reproduce the reference's latents of every step and the decoded video;
  2. times reference and restatement on the same inputs (their ratio calibrates the restatement as bench.py's cpu_baseline proxy);
  3. writes tests/golden/config1.safetensors (per-step latents + three decoded uint8 frames) for the GPU end-to-end test.
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


def config1_inputs():
    s = gen_golden.seeded
    noise = s((1, 48, 5, 16, 16), 1)
    ctx_p = s((1, 512, 4096), 2); ctx_p[:, 64:] = 0
    ctx_n = s((1, 512, 4096), 3); ctx_n[:, 128:] = 0
    return noise, ctx_p, ctx_n, s((1, 48, 1, 16, 16), 4)


def main():
    torch.set_num_threads(8)
    R = gen_golden.import_reference()
    from fairygen_amd import synthetic
    from fairygen_amd.loader import TI2V_5B_DIT_KWARGS
    from oracle import pipeline as opipe, wan_vae
    cfg = dict(TI2V_5B_DIT_KWARGS)
    t0 = time.perf_counter()
    sd = synthetic.random_state_dict(synthetic.dit_shapes(cfg), seed=1234)
    vsd = synthetic.random_state_dict(synthetic.vae_shapes(), seed=1234)
    print(f"weights: {time.perf_counter() - t0:.0f} s", flush=True)
    with torch.device("meta"):
        model = R["dit"].WanModel(**cfg)
    model.load_state_dict(sd, assign=True)
    model.freqs = R["dit"].precompute_freqs_cis_3d(cfg["dim"] // cfg["num_heads"])      # a plain attribute: built on meta above
    model.eval()
    vae = R["vae"].WanVideoVAE38()
    vae.load_state_dict(vsd, assign=True)
    vae = vae.eval()
    noise, ctx_p, ctx_n, z0 = config1_inputs()
    fn = R["pipe"].model_fn_wan_video
    steps = 4
    out, timing = {}, {}
    with torch.no_grad():
        sched = R["sched"]("Wan")
        sched.set_timesteps(steps, denoising_strength=1.0, shift=5.0)
        latents = noise.clone()
        latents[:, :, 0:1] = z0
        t0 = time.perf_counter()
        for pid, timestep in enumerate(sched.timesteps):
            t = timestep.unsqueeze(0).to(dtype=torch.bfloat16)
            posi = fn(dit=model, latents=latents, timestep=t, context=ctx_p, fuse_vae_embedding_in_latents=True)
            nega = fn(dit=model, latents=latents, timestep=t, context=ctx_n, fuse_vae_embedding_in_latents=True)
            latents = sched.step(nega + 5.0 * (posi - nega), sched.timesteps[pid], latents)
            latents[:, :, 0:1] = z0
            out[f"latents_step{pid}"] = latents.clone()
            print(f"reference step {pid}: {time.perf_counter() - t0:.0f} s", flush=True)
        timing["reference_denoise_s"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        video = vae.decode(latents, device="cpu", tiled=True, tile_size=(30, 52), tile_stride=(15, 26))
        timing["reference_decode_s"] = time.perf_counter() - t0
        print(f"reference decode: {timing['reference_decode_s']:.0f} s", flush=True)
        frames = R["base"](device="cpu", torch_dtype=torch.bfloat16).vae_output_to_video(video)
        import numpy as np
        u8 = torch.from_numpy(np.stack([np.array(f) for f in frames]))
        # ---- the oracle restatement on the same inputs
        rec = []
        t0 = time.perf_counter()
        lat_o = opipe.denoise_loop(sd, cfg, noise, ctx_p, ctx_n, steps, 5.0, 5.0, z0, record=rec)
        timing["oracle_denoise_s"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        vid_o = wan_vae.vae_decode(vsd, lat_o, True, (30, 52), (15, 26))
        timing["oracle_decode_s"] = time.perf_counter() - t0
    same_lat = all(torch.equal(r, out[f"latents_step{i}"]) for i, r in enumerate(rec))
    same_vid = torch.equal(vid_o, video)
    timing.update(oracle_equals_reference_latents=same_lat, oracle_equals_reference_video=same_vid, cores=torch.get_num_threads())
    print(json.dumps(timing, indent=1), flush=True)
    assert same_lat and same_vid, "the oracle restatement differs from the reference at full width"
    out["video_u8_frames_0_8_16"] = u8[[0, 8, 16]].contiguous()
    # the same clip evaluated in fp32 (restatement; it equals the reference in bf16 above): the yardstick that tells how far two
    # correct bf16 evaluations of this random-weight network may sit apart
    with torch.no_grad():
        sd32 = {k: v.float() for k, v in sd.items()}
        lat32 = opipe.denoise_loop(sd32, cfg, noise.float(), ctx_p.float(), ctx_n.float(), steps, 5.0, 5.0, z0.float(), dtype=torch.float32)
        del sd32
        vid32 = wan_vae.vae_decode({k: v.float() for k, v in vsd.items()}, lat32, True, (30, 52), (15, 26))
    out["latents_f32"] = lat32
    out["video_u8_f32_frames_0_8_16"] = opipe.video_to_uint8(vid32[0])[[0, 8, 16]].contiguous()
    d_lat = torch.nn.functional.cosine_similarity(lat32.flatten(), out[f"latents_step{steps - 1}"].float().flatten(), dim=0).item()
    d_u8 = (out["video_u8_f32_frames_0_8_16"].int() - out["video_u8_frames_0_8_16"].int()).abs().float().mean().item()
    timing.update(bf16_vs_f32_latents_cos=d_lat, bf16_vs_f32_frames_mean_abs_lsb=d_u8)
    print(f"reference bf16 vs fp32: latents cos {d_lat:.5f}, frames mean |diff| {d_u8:.3f} LSB", flush=True)
    gen_golden.save("config1.safetensors", out, {
        "config": "TI2V_5B_DIT_KWARGS (30 blocks, dim 3072), WanVideoVAE38() full width",
        "weights": "synthetic.random_state_dict(dit_shapes(), seed=1234) / (vae_shapes(), seed=1234), CPU generator",
        "inputs": "noise=seeded((1,48,5,16,16),1); ctx+=seeded((1,512,4096),2) rows>=64 zero; ctx-=seed 3 rows>=128 zero; "
                  "z0=seeded((1,48,1,16,16),4); 4 steps cfg 5 shift 5; decode tiled (30,52)/(15,26)",
        "timing": json.dumps(timing),
        "source": "diffsynth/pipelines/wan_video.py model_fn_wan_video + __call__ loop :283-309; WanVideoVAE38.decode; vae_output_to_video"})


if __name__ == "__main__":
    main()

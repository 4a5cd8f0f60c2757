"""Oracle: scheduler, CFG/Euler denoise loop, LoRA fuse, noise / pixel conversion (test infrastructure only).

Follows ``diffsynth/diffusion/flow_match.py:30-39,144-154``, ``diffsynth/pipelines/wan_video.py:283-325``,
``diffsynth/utils/lora/general.py:10-62``, ``diffsynth/diffusion/base_pipeline.py:128-143,171-176`` and
``merge_weights.py:19-45``.
"""
import torch

from . import wan_dit, wan_vae


def wan_sigmas(num_inference_steps, denoising_strength=1.0, shift=5.0):
    """flow_match.py:30-39 — (sigmas, timesteps) fp32 tensors of length n."""
    sigma_start = 0.0 + (1.0 - 0.0) * denoising_strength
    sigmas = torch.linspace(sigma_start, 0.0, num_inference_steps + 1)[:-1]
    sigmas = shift * sigmas / (1 + (shift - 1) * sigmas)
    return sigmas, sigmas * 1000


def euler_step(model_output, step_id, sample, sigmas):
    """flow_match.py:144-154 — x + v*(sigma' - sigma); sigma' = 0 after the last step."""
    sigma = sigmas[step_id]
    sigma_next = 0 if step_id + 1 >= len(sigmas) else sigmas[step_id + 1]
    return sample + model_output * (sigma_next - sigma)


def generate_noise(shape, seed, dtype=torch.bfloat16):
    """base_pipeline.py:171-176 — CPU fp32 randn under a seeded generator, cast to the pipeline dtype."""
    g = torch.Generator("cpu").manual_seed(seed)
    return torch.randn(shape, generator=g, device="cpu", dtype=torch.float32).to(dtype)


def denoise_loop(sd, cfg, latents, context_posi, context_nega, num_inference_steps, cfg_scale=5.0,
                 sigma_shift=5.0, first_frame_latents=None, dtype=torch.bfloat16, record=None, num_blocks=None,
                 tea_cache_l1_thresh=None, tea_cache_model_id="", denoising_strength=1.0):
    """wan_video.py:283-309 — per step: forward+, forward-, CFG combine, Euler step, re-pin frame 0.
    tea_cache_l1_thresh: one TeaCache per CFG branch (WanVideoUnit_TeaCache, :769-781)."""
    tea_p = tea_n = None
    if tea_cache_l1_thresh is not None:
        tea_p = wan_dit.TeaCache(num_inference_steps, tea_cache_l1_thresh, tea_cache_model_id)
        tea_n = wan_dit.TeaCache(num_inference_steps, tea_cache_l1_thresh, tea_cache_model_id)
    sigmas, timesteps = wan_sigmas(num_inference_steps, denoising_strength, shift=sigma_shift)
    fuse = first_frame_latents is not None
    if fuse:
        latents = latents.clone()
        latents[:, :, 0:1] = first_frame_latents
    for i, ts in enumerate(timesteps):
        t = ts.unsqueeze(0).to(dtype=dtype)                      # wan_video.py:293 (bf16 rounding)
        posi = wan_dit.model_fn(sd, cfg, latents, t, context_posi, fuse, num_blocks, tea_cache=tea_p)
        if cfg_scale != 1.0:
            nega = wan_dit.model_fn(sd, cfg, latents, t, context_nega, fuse, num_blocks, tea_cache=tea_n)
            pred = nega + cfg_scale * (posi - nega)
        else:
            pred = posi
        latents = euler_step(pred, i, latents, sigmas)
        if fuse:
            latents[:, :, 0:1] = first_frame_latents
        if record is not None:
            record.append(latents.clone())
    return latents


def add_noise(original_samples, noise, sigma):
    """flow_match.py:164-170 — (1 - sigma) x0 + sigma eps at the schedule's first sigma (video-to-video start)."""
    return (1 - sigma) * original_samples + sigma * noise


def video_to_uint8(video):
    """base_pipeline.py:128-143 — (C,T,H,W) in [-1,1] -> (T,H,W,C) uint8 by truncation."""
    x = video.permute(1, 2, 3, 0)
    return ((x - (-1)) * (255 / 2)).clip(0, 255).to(device="cpu", dtype=torch.uint8)


# ------------------------------------------------------------------------------------------- LoRA
def lora_name_map(lora_sd):
    """general.py:10-30 — target module name -> (B key, A key); accepts lora_up/down, adapter names,
    and a leading 'diffusion_model.'."""
    out = {}
    for key in lora_sd:
        a_tag, b_tag = ("lora_down", "lora_up") if ".lora_up." in key else ("lora_A", "lora_B")
        if b_tag not in key:
            continue
        parts = key.split(".")
        pos = parts.index(b_tag)
        if len(parts) > pos + 2:
            parts.pop(pos + 1)          # adapter name, e.g. "default"
        parts.pop(pos)
        if parts[0] == "diffusion_model":
            parts.pop(0)
        parts.pop(-1)                   # "weight"
        out[".".join(parts)] = (key, key.replace(b_tag, a_tag))
    return out


def fuse_lora(sd, lora_sd, alpha=1.0):
    """general.py:44-62 — W <- W + alpha * (B @ A), in the dtype of W; returns #fused."""
    n = 0
    for name, (kb, ka) in lora_name_map(lora_sd).items():
        wk = name + ".weight"
        if wk not in sd:
            continue
        up, down = lora_sd[kb].to(sd[wk].dtype), lora_sd[ka].to(sd[wk].dtype)
        if up.dim() == 4:
            delta = alpha * torch.mm(up.squeeze(3).squeeze(2), down.squeeze(3).squeeze(2)).unsqueeze(2).unsqueeze(3)
        else:
            delta = alpha * torch.mm(up, down)
        sd[wk] = sd[wk] + delta
        n += 1
    return n


def hot_lora_linear(x, weight, bias, adapters):
    """core/vram/layers.py:410-436 — unfused (hot-loaded) adapters: linear(x), then out + x @ A^T @ B^T per adapter, left to
    right in the tensors' dtype; `adapters` = [(alpha * A, B), ...] as base_pipeline.py:258-259 stores them."""
    out = torch.nn.functional.linear(x, weight, bias)
    for a, b in adapters:
        out = out + x @ a.T @ b.T
    return out


def merge_stage_loras(stage1, stage2):
    """merge_weights.py:19-45 — merged = {A1, B1 + B2}."""
    merged = {}
    for k, v in stage1.items():
        if "lora_A" in k:
            merged[k] = v
        elif "lora_B" in k:
            if k.endswith(".lora_B.default.weight"):
                k2 = k.replace(".lora_B.default.weight", ".lora_B2.weight")
            else:
                k2 = k.replace("lora_B", "lora_B2").replace(".default", "")
            merged[k] = v + stage2[k2] if k2 in stage2 else v
    return merged


def generate_clip(dit_sd, dit_cfg, vae_sd, noise, context_posi, context_nega, num_inference_steps,
                  cfg_scale=5.0, sigma_shift=5.0, first_frame_latents=None, tiled=True,
                  tile_size=(30, 52), tile_stride=(15, 26)):
    """wan_video.py:247-325 from the scheduler set-up to the decoded (1,3,F,H,W) tensor."""
    latents = denoise_loop(dit_sd, dit_cfg, noise, context_posi, context_nega, num_inference_steps,
                           cfg_scale, sigma_shift, first_frame_latents, dtype=noise.dtype)
    return latents, wan_vae.vae_decode(vae_sd, latents, tiled, tile_size, tile_stride)

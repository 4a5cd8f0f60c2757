"""Generate tests/golden/*.safetensors by RUNNING THE REFERENCE's own Python (build container only).

    python oracle/gen_golden.py            # needs /root/reference; writes tests/golden/

The reference (CloudEngineHub/FairyGen, animation/diffsynth) ships no tests or golden vectors for the
Wan path, so these vectors — outputs of the unmodified reference modules on seeded inputs
and on the deterministic synthetic weights of ``fairygen_amd.synthetic`` — are what pins the oracle.
Import recipe = SURVEY.md §8(c): inert stubs for the absent, non-arithmetic dependencies
(modelscope, ftfy, peft, imageio, torchvision), then a normal import of ``diffsynth``.
Nothing of the reference is copied: only tensors (inputs' seeds + expected outputs) are stored.
"""
import ast
import os
import sys
import tempfile
import types

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/animation"
OUT = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, REPO)


def import_reference():
    import transformers  # noqa: F401  (let its optional-dependency probing finish first)

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    def _no_download(*a, **k):
        raise RuntimeError("offline")

    stub("modelscope", snapshot_download=_no_download, dataset_snapshot_download=_no_download)
    stub("ftfy", fix_text=lambda s: s)
    stub("peft", LoraConfig=object, inject_adapter_in_model=_no_download)
    stub("imageio").v3 = stub("imageio.v3")
    tv = stub("torchvision")
    tv.transforms = stub("torchvision.transforms")
    tv.transforms.functional = stub("torchvision.transforms.functional")
    sys.path.insert(0, REF)
    from diffsynth.pipelines import wan_video as ref_pipe
    from diffsynth.models import wan_video_dit as ref_dit, wan_video_vae as ref_vae
    from diffsynth.diffusion.flow_match import FlowMatchScheduler
    from diffsynth.diffusion.base_pipeline import BasePipeline
    from diffsynth.utils.lora.general import GeneralLoRALoader
    from diffsynth.core.loader.file import hash_state_dict_keys
    from diffsynth.models import wan_video_text_encoder as ref_text
    globals()["_REF_TEXT"] = ref_text
    return dict(pipe=ref_pipe, dit=ref_dit, vae=ref_vae, sched=FlowMatchScheduler, base=BasePipeline,
                lora=GeneralLoRALoader, hash=hash_state_dict_keys)


def seeded(shape, seed, dtype=torch.bfloat16, scale=1.0):
    g = torch.Generator("cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g, dtype=torch.float32) * scale).to(dtype)


def save(name, tensors, meta):
    from safetensors.torch import save_file
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name)
    save_file({k: v.contiguous() for k, v in tensors.items()}, path, metadata={k: str(v) for k, v in meta.items()})
    print(f"wrote {path}: {sum(v.numel() * v.element_size() for v in tensors.values()) / 1e3:.1f} kB")


def section_serving(R):
    """7. serving glue next to the loop: TeaCache step skipping (8-step CFG loop on the tiny DiT)."""
    from fairygen_amd import synthetic
    ref_dit, ref_pipe = R["dit"], R["pipe"]
    kw = synthetic.TINY_DIT_KWARGS
    sd = synthetic.random_state_dict(synthetic.dit_shapes(kw), seed=1234)
    model = ref_dit.WanModel(**kw).to(torch.bfloat16).eval()
    model.load_state_dict(sd)
    lat = seeded((1, 48, 3, 8, 8), 1)
    ctx_p = seeded((1, 16, 128), 2); ctx_p[:, 10:] = 0
    ctx_n = seeded((1, 16, 128), 3); ctx_n[:, 12:] = 0
    z0 = seeded((1, 48, 1, 8, 8), 4)
    fn = ref_pipe.model_fn_wan_video
    steps, thresh, model_id = 8, 25.0, "Wan2.1-I2V-14B-720P"      # synthetic weights: rescaled distances 8-21 per step -> a mix of skipped and computed steps
    thresh_t2v_skip = 100.0      # T2V mode: rescaled distances 42-96 per step, so 25 skips nothing there; 100 skips steps 1, 2 and 4
    out = {}
    for mode, fuse, thr in (("ti2v", True, thresh), ("t2v", False, thresh), ("t2v_skip", False, thresh_t2v_skip)):
        tea = [ref_pipe.TeaCache(steps, rel_l1_thresh=thr, model_id=model_id) for _ in range(2)]
        sched = R["sched"]("Wan")
        sched.set_timesteps(steps, denoising_strength=1.0, shift=5.0)
        latents = lat.clone()
        if fuse:
            latents[:, :, 0:1] = z0
        skipped = []
        with torch.no_grad():
            for pid, timestep in enumerate(sched.timesteps):
                t = timestep.unsqueeze(0).to(dtype=torch.bfloat16)
                had = [tc.previous_residual is not None and tc.previous_hidden_states is None for tc in tea]
                posi = fn(dit=model, latents=latents, timestep=t, context=ctx_p, fuse_vae_embedding_in_latents=fuse, tea_cache=tea[0])
                nega = fn(dit=model, latents=latents, timestep=t, context=ctx_n, fuse_vae_embedding_in_latents=fuse, tea_cache=tea[1])
                # a skipped step leaves previous_hidden_states None (check() only clones x when it computes)
                skipped.append([int(tc.previous_hidden_states is None and h and tc.accumulated_rel_l1_distance != 0) for tc, h in zip(tea, had)])
                pred = nega + 5.0 * (posi - nega)
                latents = sched.step(pred, sched.timesteps[pid], latents)
                if fuse:
                    latents[:, :, 0:1] = z0
                out[f"{mode}_step{pid}"] = latents.clone()
                out[f"{mode}_acc{pid}"] = torch.tensor([float(tc.accumulated_rel_l1_distance) for tc in tea], dtype=torch.float64)
        out[f"{mode}_skipped"] = torch.tensor(skipped)
        print(mode, "skipped (posi, nega) per step:", skipped)
    # sliding-window mode of model_fn_wan_video (TemporalTiler_BCTHW): 7 latent frames, windows of 4 with stride 2
    lat7 = seeded((1, 48, 7, 8, 8), 98)
    with torch.no_grad():
        out["sliding_window_out"] = fn(dit=model, latents=lat7, timestep=torch.tensor([700.0]).to(torch.bfloat16), context=ctx_p,
                                       fuse_vae_embedding_in_latents=True, sliding_window_size=4, sliding_window_stride=2)
    # hot-loaded LoRA: AutoWrappedLinear.forward = linear_forward, then lora_forward (core/vram/layers.py:410-436) with the
    # lists load_lora fills (alpha folded into A, base_pipeline.py:258); two stacked adapters
    from diffsynth.core.vram.layers import AutoWrappedLinear
    x = seeded((1, 40, 256), 91)
    w, b = seeded((384, 256), 92, scale=0.05), seeded((384,), 93, scale=0.1)
    a1, b1 = seeded((4, 256), 94, scale=0.05), seeded((384, 4), 95, scale=0.05)
    a2, b2 = seeded((8, 256), 96, scale=0.05), seeded((384, 8), 97, scale=0.05)
    stub = types.SimpleNamespace(lora_merger=None, lora_A_weights=[a1 * 0.5, a2 * 2], lora_B_weights=[b1, b2])
    out["hot_lora_out"] = AutoWrappedLinear.lora_forward(stub, x, torch.nn.functional.linear(x, w, b))
    try:
        ref_pipe.TeaCache(4, 0.1, "Wan2.2-TI2V-5B")
        raise AssertionError("expected ValueError")
    except ValueError:
        pass
    save("serving.safetensors", out, {
        "config": str(kw), "weights": "synthetic.random_state_dict(dit_shapes(TINY_DIT_KWARGS), seed=1234)",
        "inputs": f"as dit_tiny.safetensors; {steps} steps cfg 5 shift 5; TeaCache(rel_l1_thresh={thresh}, model_id={model_id!r}) per CFG branch; "
                  f"t2v: fuse_vae_embedding_in_latents=False (no first-frame pin); t2v_skip: the same with rel_l1_thresh={thresh_t2v_skip}; sliding window: latents=seeded((1,48,7,8,8),98) t=bf16(700) ctx+ size 4 stride 2; hot LoRA: x=seeded((1,40,256),91) w=seed 92 x0.05 (384,256) "
                  "b=seed 93 x0.1; adapters (A1 seed 94 (4,256), B1 seed 95, alpha 0.5), (A2 seed 96 (8,256), B2 seed 97, alpha 2), all x0.05",
        "source": "diffsynth/pipelines/wan_video.py TeaCache :1016-1065, model_fn_wan_video :1297-1300,1316-1317,1375-1376; "
                  "core/vram/layers.py AutoWrappedLinear.lora_forward :417-428; base_pipeline.py:249-262; TemporalTiler_BCTHW :1069-1118"})


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    R = import_reference()
    from fairygen_amd import synthetic
    if "--only-serving" in sys.argv:
        return section_serving(R)

    # ------------------------------------------------------------------ 1. DiT primitives
    ref_dit = R["dit"]
    out = {}
    x = seeded((1, 24, 256), 11)
    freqs3 = ref_dit.precompute_freqs_cis_3d(128)
    f, h, w = 2, 3, 4
    table = torch.cat([freqs3[0][:f].view(f, 1, 1, -1).expand(f, h, w, -1),
                       freqs3[1][:h].view(1, h, 1, -1).expand(f, h, w, -1),
                       freqs3[2][:w].view(1, 1, w, -1).expand(f, h, w, -1)], dim=-1).reshape(f * h * w, 1, -1)
    out["rope_out"] = ref_dit.rope_apply(x, table, 2)
    out["rope_table_real"], out["rope_table_imag"] = table.real.contiguous(), table.imag.contiguous()
    out["sinusoid_bf16"] = ref_dit.sinusoidal_embedding_1d(256, torch.tensor([0.0, 996.0, 92.5], dtype=torch.bfloat16))
    out["sinusoid_f32"] = ref_dit.sinusoidal_embedding_1d(256, torch.tensor([0.0, 995.9, 92.59], dtype=torch.float32))
    rms = ref_dit.RMSNorm(256, eps=1e-6)
    rms.weight.data = (1 + 0.1 * seeded((256,), 12, torch.float32)).to(torch.bfloat16)
    out["rmsnorm_out"] = rms(x)
    out["modulate_out"] = ref_dit.modulate(torch.nn.functional.layer_norm(x, (256,), eps=1e-6), seeded((1, 1, 256), 13), seeded((1, 1, 256), 14))
    q, k, v = seeded((1, 80, 256), 15), seeded((1, 50, 256), 16), seeded((1, 50, 256), 17)
    out["attn_out_bf16"] = ref_dit.flash_attention(q, k, v, num_heads=2)
    out["attn_out_f32"] = ref_dit.flash_attention(q.float(), k.float(), v.float(), num_heads=2)
    gate = ref_dit.GateModule()
    out["gate_out"] = gate(x, seeded((1, 1, 256), 18), seeded((1, 24, 256), 19))
    save("dit_primitives.safetensors", out, {
        "inputs": "x=seeded((1,24,256),11) grid=(2,3,4) heads=2; rms weight=1+0.1*seeded((256,),12,f32)->bf16; "
                  "modulate shift=seed13 scale=seed14; q=seed15 (1,80,256) k=seed16 v=seed17 (1,50,256); gate=seed18 res=seed19",
        "source": "diffsynth/models/wan_video_dit.py rope_apply/sinusoidal_embedding_1d/RMSNorm/modulate/flash_attention/GateModule"})

    # ------------------------------------------------------------------ 2. tiny DiT forward (both timestep modes) + 4-step loop
    kw = synthetic.TINY_DIT_KWARGS
    sd = synthetic.random_state_dict(synthetic.dit_shapes(kw), seed=1234)
    model = ref_dit.WanModel(**kw).to(torch.bfloat16).eval()
    model.load_state_dict(sd)
    lat = seeded((1, 48, 3, 8, 8), 1)
    ctx_p = seeded((1, 16, 128), 2); ctx_p[:, 10:] = 0
    ctx_n = seeded((1, 16, 128), 3); ctx_n[:, 12:] = 0
    z0 = seeded((1, 48, 1, 8, 8), 4)
    ts = torch.tensor([995.9]).to(torch.bfloat16)
    fn = R["pipe"].model_fn_wan_video
    out = {}
    with torch.no_grad():
        out["ti2v_bf16"] = fn(dit=model, latents=lat, timestep=ts, context=ctx_p, fuse_vae_embedding_in_latents=True)
        out["t2v_bf16"] = fn(dit=model, latents=lat, timestep=ts, context=ctx_p, fuse_vae_embedding_in_latents=False)
        m32 = ref_dit.WanModel(**kw).float().eval()
        m32.load_state_dict({k_: v_.float() for k_, v_ in sd.items()})
        out["ti2v_f32"] = fn(dit=m32, latents=lat.float(), timestep=ts.float(), context=ctx_p.float(), fuse_vae_embedding_in_latents=True)
        # the denoise loop of WanVideoPipeline.__call__ (wan_video.py:283-309), 4 steps, cfg 5, shift 5
        sched = R["sched"]("Wan")
        sched.set_timesteps(4, denoising_strength=1.0, shift=5.0)
        latents = lat.clone()
        latents[:, :, 0:1] = z0
        for pid, timestep in enumerate(sched.timesteps):
            t = timestep.unsqueeze(0).to(dtype=torch.bfloat16)
            posi = fn(dit=model, latents=latents, timestep=t, context=ctx_p, fuse_vae_embedding_in_latents=True)
            nega = fn(dit=model, latents=latents, timestep=t, context=ctx_n, fuse_vae_embedding_in_latents=True)
            pred = nega + 5.0 * (posi - nega)
            latents = sched.step(pred, sched.timesteps[pid], latents)
            latents[:, :, 0:1] = z0
            out[f"loop_step{pid}"] = latents.clone()
    assert R["hash"](model.state_dict()) == R["hash"](sd)
    save("dit_tiny.safetensors", out, {
        "config": str(kw), "weights": "synthetic.random_state_dict(dit_shapes(TINY_DIT_KWARGS), seed=1234)",
        "inputs": "latents=seeded((1,48,3,8,8),1); ctx+=seeded((1,16,128),2) rows>=10 zero; ctx-=seed 3 rows>=12 zero; "
                  "z0=seeded((1,48,1,8,8),4); timestep=bf16(995.9); loop: 4 steps cfg 5 shift 5",
        "source": "diffsynth/pipelines/wan_video.py model_fn_wan_video + __call__ loop :283-309; FlowMatchScheduler"})

    # ------------------------------------------------------------------ 3. scheduler tables
    out = {}
    for n in (4, 30, 50):
        s = R["sched"]("Wan")
        s.set_timesteps(n, denoising_strength=1.0, shift=5.0)
        out[f"sigmas_{n}"], out[f"timesteps_{n}"] = s.sigmas, s.timesteps
    s = R["sched"]("Wan")
    s.set_timesteps(4, shift=5.0)
    xs, vs = seeded((1, 4, 2, 3, 3), 21), seeded((1, 4, 2, 3, 3), 22)
    for i in range(4):
        out[f"step_{i}"] = s.step(vs, s.timesteps[i], xs)
    save("scheduler.safetensors", out, {"inputs": "sample=seeded((1,4,2,3,3),21) model_output=seed 22; shift 5",
                                        "source": "diffsynth/diffusion/flow_match.py:30-39,144-154"})

    # ------------------------------------------------------------------ 4. LoRA fuse + merge_weights
    lora = synthetic.random_lora(synthetic.dit_shapes(kw), rank=4, seed=4321)
    model2 = ref_dit.WanModel(**kw).to(torch.bfloat16).eval()
    model2.load_state_dict(sd)
    loader = R["lora"](device="cpu", torch_dtype=torch.bfloat16)
    loader.fuse_lora_to_base_model(model2, loader.convert_state_dict(lora), alpha=0.5)
    fused = model2.state_dict()
    out = {k_: fused[k_] for k_ in ("blocks.0.self_attn.q.weight", "blocks.1.cross_attn.v.weight", "blocks.1.ffn.2.weight",
                                    "blocks.0.ffn.0.weight", "blocks.0.self_attn.q.bias")}
    src = open("/root/reference/animation/merge_weights.py").read()
    fn_node = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "merge_lora_weights"][0]
    ns = {}
    import safetensors.torch
    exec(compile(ast.Module(body=[fn_node], type_ignores=[]), "merge_weights.py", "exec"), {"safetensors": safetensors, "torch": torch}, ns)
    stage1 = lora
    stage2 = {k_.replace(".lora_B.default.weight", ".lora_B2.weight"): seeded(v_.shape, 77 + i, scale=0.02)
              for i, (k_, v_) in enumerate(sorted(lora.items())) if ".lora_B." in k_}
    with tempfile.TemporaryDirectory() as td:
        safetensors.torch.save_file(stage1, td + "/s1.safetensors")
        safetensors.torch.save_file(stage2, td + "/s2.safetensors")
        ns["merge_lora_weights"](td + "/s1.safetensors", td + "/s2.safetensors", td + "/m.safetensors")
        merged = safetensors.torch.load_file(td + "/m.safetensors")
    out["merged.blocks.0.self_attn.q.lora_B"] = merged["blocks.0.self_attn.q.lora_B.default.weight"]
    out["merged.blocks.1.ffn.0.lora_A"] = merged["blocks.1.ffn.0.lora_A.default.weight"]
    out["merged_num_keys"] = torch.tensor([len(merged)])
    save("lora.safetensors", out, {
        "inputs": "lora=synthetic.random_lora(dit_shapes(TINY), rank=4, seed=4321), alpha=0.5 on the tiny DiT (seed 1234); "
                  "stage2 B2[i]=seeded(shape, 77+i, scale=0.02) over sorted lora_B keys",
        "source": "diffsynth/utils/lora/general.py:44-62; merge_weights.py:19-45"})

    # ------------------------------------------------------------------ 5. VAE38 decode (tiny decoder), untiled + tiled
    ref_vae = R["vae"]
    dim, dec_dim = 32, 32
    vsd = synthetic.random_state_dict(synthetic.vae_shapes(dec_dim=dec_dim, dim=dim), seed=1234)
    inner = ref_vae.VideoVAE38_(dim=dim, z_dim=48, dec_dim=dec_dim).eval().requires_grad_(False)
    wrap = ref_vae.WanVideoVAE38.__new__(ref_vae.WanVideoVAE38)
    torch.nn.Module.__init__(wrap)
    # (the full WanVideoVAE38() would build 700 M parameters; the wrapper is assembled by hand around the tiny model)
    from oracle.wan_vae import VAE38_MEAN, VAE38_STD
    wrap.mean, wrap.std = torch.tensor(VAE38_MEAN), torch.tensor(VAE38_STD)
    wrap.scale = [wrap.mean, 1.0 / wrap.std]
    wrap.model, wrap.upsampling_factor, wrap.z_dim = inner, 16, 48
    wrap = wrap.to(torch.bfloat16)
    wrap.load_state_dict(vsd)
    z = seeded((1, 48, 3, 4, 6), 31)
    out = {}
    with torch.no_grad():
        out["decode_bf16"] = wrap.decode(z, device="cpu", tiled=False)
        out["tiled_bf16"] = wrap.decode(z, device="cpu", tiled=True, tile_size=(3, 4), tile_stride=(2, 2))
        w32 = wrap.float()
        out["decode_f32"] = w32.decode(z.float(), device="cpu", tiled=False)
        # encoder: first-frame image (the TI2V conditioning path), a 9-frame video (chunked encode), tiled image
        img = seeded((3, 1, 64, 96), 32, scale=0.5).clamp(-1, 1)
        vid = seeded((3, 9, 64, 96), 33, scale=0.5).clamp(-1, 1)
        wb = wrap.to(torch.bfloat16)
        out["encode_image_bf16"] = wb.encode([img], device="cpu")
        out["encode_video_bf16"] = wb.encode([vid], device="cpu")
        out["encode_image_tiled_bf16"] = wb.encode([img], device="cpu", tiled=True, tile_size=(3, 4), tile_stride=(2, 2))
        out["encode_image_f32"] = wb.float().encode([img.float()], device="cpu")
    # the class constants the wrapper was built from must be the reference's
    ref_full_src = open("/root/reference/animation/diffsynth/models/wan_video_vae.py").read()
    assert "-0.2289, -0.0052, -0.1323" in ref_full_src and "0.4765, 1.0364, 0.4514" in ref_full_src
    save("vae_tiny.safetensors", out, {
        "config": f"VideoVAE38_(dim={dim}, z_dim=48, dec_dim={dec_dim})",
        "weights": f"synthetic.random_state_dict(vae_shapes(dec_dim={dec_dim}, dim={dim}), seed=1234)",
        "inputs": "z=seeded((1,48,3,4,6),31); tiled: tile_size=(3,4) tile_stride=(2,2); encode: img=seeded((3,1,64,96),32,scale=0.5).clamp(-1,1), "
                  "vid=seeded((3,9,64,96),33,scale=0.5).clamp(-1,1)",
        "source": "diffsynth/models/wan_video_vae.py WanVideoVAE.decode/tiled_decode :1103-1152,1235-1247; VideoVAE38_.decode :1326-1351"})

    # ------------------------------------------------------------------ 5b. umT5 text encoder (tiny config) + padded-row zeroing
    tkw = synthetic.TINY_TEXT_KWARGS
    tsd = {k_: (v_ * 3 if v_.dim() == 2 else v_) for k_, v_ in
           synthetic.random_state_dict(synthetic.text_encoder_shapes(tkw), seed=1234).items()}
    tenc = _REF_TEXT.WanTextEncoder(**tkw).to(torch.bfloat16).eval()
    tenc.load_state_dict(tsd)
    g_ids = torch.Generator("cpu").manual_seed(51)
    ids = torch.randint(0, tkw["vocab"], (1, 24), generator=g_ids)
    mask = torch.zeros((1, 24), dtype=torch.long); mask[:, :17] = 1
    out = {"ids": ids, "mask": mask}
    with torch.no_grad():
        emb = tenc(ids, mask)
        out["encoder_bf16"] = emb.clone()
        for v_ in mask.gt(0).sum(dim=1).long():          # WanVideoUnit_PromptEmbedder.encode_prompt :408-411
            emb[:, v_:] = 0
        out["prompt_emb_bf16"] = emb
        t32 = _REF_TEXT.WanTextEncoder(**tkw).float().eval()
        t32.load_state_dict({k_: v_.float() for k_, v_ in tsd.items()})
        out["encoder_f32"] = t32(ids, mask)
    out["clean"] = torch.tensor([ord(ch) for ch in _REF_TEXT.whitespace_clean(_REF_TEXT.basic_clean("  a &amp;amp; b \n\t c  "))])
    save("text_tiny.safetensors", out, {
        "config": str(tkw), "weights": "synthetic.random_state_dict(text_encoder_shapes(TINY_TEXT_KWARGS), seed=1234), 2-D tensors x3",
        "inputs": "ids=randint(0,100,(1,24)) seed 51; mask first 17 ones",
        "source": "diffsynth/models/wan_video_text_encoder.py WanTextEncoder.forward; pipelines/wan_video.py:404-412"})

    # ------------------------------------------------------------------ 6. pixel / noise conventions
    base = R["base"](device="cpu", torch_dtype=torch.bfloat16)
    out = {"noise_seed1": base.generate_noise((1, 48, 2, 4, 4), seed=1, rand_device="cpu")}
    vid = (seeded((1, 3, 2, 8, 8), 41, scale=0.7)).clamp(-1.2, 1.2)
    frames = base.vae_output_to_video(vid)
    import numpy as np
    out["uint8_frames"] = torch.from_numpy(np.stack([np.array(f) for f in frames]))
    save("pixels.safetensors", out, {"inputs": "noise shape (1,48,2,4,4) seed 1; video=seeded((1,3,2,8,8),41,scale=0.7).clamp(-1.2,1.2)",
                                     "source": "diffsynth/diffusion/base_pipeline.py:128-143,171-176"})

    section_serving(R)


if __name__ == "__main__":
    main()

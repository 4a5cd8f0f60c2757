"""CPU-side tests: C-ABI surface, model identification, scheduler / LoRA / pipeline plumbing mirrors, and the
loud-failure contract (no CPU fallback).  No kernel is launched here."""
import ctypes
import os
import re

import pytest
import torch

from conftest import REPO, seeded
from fairygen_amd import ModelConfig, WanVideoPipeline, hip, loader, synthetic
from fairygen_amd.flow_match import FlowMatchScheduler
from fairygen_amd.lora import GeneralLoRALoader, merge_lora_weights
from fairygen_amd.wan_video_dit import WanModel
from fairygen_amd.wan_video_vae import WanVideoVAE38
from oracle import wan_vae


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(REPO, "include", "fairygen_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(fg_[a-z0-9_]+)\s*\(", header)))
    assert declared == hip.EXPORTED_SYMBOLS
    lib = ctypes.CDLL(hip.library_path())
    for name in declared:
        assert hasattr(lib, name), name
    assert hip.load().fg_version() == hip.ABI_VERSION


def test_no_cpu_fallback():
    q = seeded((1, 8, 256), 1)
    with pytest.raises(hip.HipLibraryError, match="no CPU fallback"):
        hip.attention(q, q, q, 2)
    with pytest.raises(hip.HipLibraryError):
        hip.activation(q, "silu")
    with pytest.raises(hip.HipLibraryError):
        hip.vae_rmsnorm_silu(q, q[0, 0], True)
    w, b = seeded((256, 256), 2), seeded((256,), 3)
    with pytest.raises(hip.HipLibraryError, match="no CPU fallback"):
        hip.gemm_epilogue(q, w, b)
    with pytest.raises(hip.HipLibraryError):
        hip.ln_modulate_fp8(q, None, 0, 1, 1e-6)


def test_gemm_entry_point_validation_and_backend_policy():
    """fg_gemm_epilogue_bf16's argument checks run on the host (no launch); which DiT Linears go to it is shape policy."""
    import ctypes
    from fairygen_amd import wan_video_dit as wd
    lib = hip.load()
    p16 = ctypes.c_void_p(16)
    call = lambda M, N, K, mode=0, gate=None, rows=1, ld=0, ws=None: lib.fg_gemm_epilogue_bf16(      # noqa: E731
        p16, K, p16, p16, p16, N, M, N, K, mode, gate, rows, ld, 0, ws, None)
    assert call(512, 250, 256) == -1 and b"N % 256" in lib.fg_last_error()
    assert call(512, 256, 192) == -1 and b"K % 128" in lib.fg_last_error()
    assert call(512, 576, 384) == -1 and b"N % 256" in lib.fg_last_error()      # the 192-column tiling of round 2 is gone
    assert call(512, 256, 256, mode=1) == -1 and b"mode must be" in lib.fg_last_error()
    assert call(512, 256, 256, mode=5) == -1 and b"mode must be" in lib.fg_last_error()
    assert call(512, 256, 256, mode=2) == -1 and b"gate table" in lib.fg_last_error()
    assert call(512, 256, 256, ws=ctypes.c_void_p(8)) == -1 and b"workspace" in lib.fg_last_error()
    assert lib.fg_gemm_workspace_bytes(27280, 3072, 14336) == 256 * 256 * 256 * 4      # one fp32 tile per CU (256 assumed without a device)
    assert lib.fg_gemm_workspace_bytes(600, 576, 384) == 0
    # the e4m3 form: same checks, K in 256s, a scale pointer
    call8 = lambda M, N, K, scale=p16, mode=0: lib.fg_gemm_fp8_bf16(p16, K, scale, p16, p16, p16, N, M, N, K, mode, None, 1, 0, 0, None, None)      # noqa: E731
    assert call8(512, 256, 384) == -1 and b"K % 256" in lib.fg_last_error()
    assert call8(512, 256, 512, scale=None) == -1 and b"null pointer" in lib.fg_last_error()
    assert call8(512, 256, 512, mode=7) == -1 and b"mode must be" in lib.fg_last_error()
    # the policy: every shape the persistent kernel takes with at least 64 tiles, shard sizes included
    assert wd.GEMM_BACKEND in ("fused", "fused-ffn2", "all", "lib")
    if wd.GEMM_BACKEND != "lib":
        assert wd.own_gemm_ok(27280, 3072, 3072) and wd.own_gemm_ok(13640, 3072, 3072) and wd.own_gemm_ok(27280, 3072, 14336)
        assert wd.own_gemm_ok(6820, 3072, 3072) and wd.own_gemm_ok(3410, 3072, 3072) and wd.own_gemm_ok(3410, 3072, 14336)      # 1/4 and 1/8 token shards
        assert not wd.own_gemm_ok(320, 256, 256)               # the tiny test models: a handful of tiles, left to the library
        assert not wd.own_gemm_ok(27280, 128, 128) and not wd.own_gemm_ok(27280, 3072, 3000)


def test_attention_scale_fold_is_host_arithmetic():
    """The self-attention scale split: 1/sqrt(d) * log2(e) = 2^e * f with f in [1/sqrt2, sqrt2); scale' = 2^e / log2(e) is what the kernel
    gets (its pre-multiplied form is exact for it), f goes into q's RoPE table; WanModel.attn_scale() turns the fold off whenever q must
    stay as the reference rounds it (fp64 RoPE mode, fold_attn_scale = False, a replaced AttentionModule)."""
    import math
    from fairygen_amd.wan_video_dit import AttentionModule
    scale, f = hip.pow2_softmax_scale(128)
    assert abs(scale * math.log2(math.e) - 2.0 ** -3) < 1e-15 and abs(f - 1.0201394465) < 1e-9
    assert abs(scale * f - 128 ** -0.5) < 1e-15
    for d in (64, 96, 128, 256):
        sc, ff = hip.pow2_softmax_scale(d)
        assert math.log2(sc * math.log2(math.e)).is_integer() and 2 ** -0.5 <= ff < 2 ** 0.5 and abs(sc * ff - d ** -0.5) < 1e-15
    m = WanModel(**synthetic.TINY_DIT_KWARGS)
    assert m.attn_scale() == (scale, f)
    tk, tq = m.rope_tables(2, 3, 4, "cpu")
    assert tk.dtype == torch.float32 and tk.shape == (24, 64, 2) and torch.allclose(tq, tk * f, rtol=1e-6) and not torch.equal(tq, tk)
    m.fold_attn_scale = False
    assert m.attn_scale() == (None, 1.0) and m.rope_tables(2, 3, 4, "cpu")[0] is m.rope_tables(2, 3, 4, "cpu")[1]
    m.fold_attn_scale, m.rope_mode = True, "f64"
    assert m.attn_scale() == (None, 1.0) and m.rope_tables(2, 3, 4, "cpu")[0].dtype == torch.float64
    m.rope_mode = "f32"

    class Plugged(AttentionModule):
        pass
    m.blocks[0].self_attn.attn = Plugged(m.num_heads)
    assert m.attn_scale() == (None, 1.0)          # a user-supplied attention module gets q exactly as the reference computes it


def test_argument_validation_through_the_c_abi():
    lib = hip.load()
    rc = lib.fg_attn_fwd_bf16(None, 0, None, 0, None, 0, None, 1, 1, 1, 1, 128, 1.0, None, 0, None)
    assert rc == -1 and b"null pointer" in lib.fg_last_error()
    rc = lib.fg_conv3d_cl_bf16(ctypes.c_void_p(16), ctypes.c_void_p(16), ctypes.c_void_p(16), None, ctypes.c_void_p(16),
                               1, 4, 4, 8, 8, 2, 3, 0, 0, None)
    assert rc == -1 and b"kt and ks" in lib.fg_last_error()
    assert lib.fg_conv_packed_bytes(12, 256, 3, 3, 3) == 27 * 128 * 256 * 2
    assert lib.fg_conv_packed_bytes(1024, 48, 3, 3, 3) == 27 * 1024 * 64 * 2


def test_attention_split_choice_is_host_logic():
    """The split-KV decomposition is chosen on the host from the shape alone (256 workgroup slots assumed without a GPU)."""
    lib = hip.load()
    R, S = ctypes.c_int(), ctypes.c_int()
    def choice(nq, nkv, ws=1 << 40):
        assert lib.fg_attn_split_choice(1, nq, nkv, 24, ws, ctypes.byref(R), ctypes.byref(S)) == 0
        return R.value, S.value
    assert choice(27280, 27280) == (1, 16)          # 2568 workgroups = 10.03 rounds: cut each head's tail q-block
    assert choice(3410, 27280)[0] == 14             # a 1/8 token shard: every q-block is cut into KV ranges
    assert choice(27280, 512) == (0, 1)             # cross-attention: 8 KV tiles, nothing to balance
    assert choice(27280, 27280, ws=0) == (0, 1)     # no workspace -> no split
    assert lib.fg_attn_workspace_bytes(1, 27280, 27280, 24) == 24 * 16 * (256 * 128 * 4 + 256 * 2 * 4)
    assert lib.fg_attn_workspace_bytes(1, 27280, 512, 24) == 0


def test_key_hashes_match_the_reference_table():
    # configs/model_configs.py:289-302 of the reference
    dit = {k: list(v) for k, v in synthetic.dit_shapes().items()}
    vae = {k: list(v) for k, v in synthetic.vae_shapes(with_prefix=False).items()}
    assert len(dit) == 825
    assert loader.hash_keys_dict(dit) == "1f5ab7703c6fc803fdded85ff040c316"
    assert loader.hash_keys_dict(vae) == "e1de6c02cdac79f8b739f4d3698cd216"


def test_model_pool_loads_by_hash_and_rejects_unknown(tmp_path):
    cfg = synthetic.TINY_DIT_KWARGS
    sd = synthetic.random_state_dict(synthetic.dit_shapes(cfg), seed=1234)
    path = synthetic.save_checkpoint(sd, str(tmp_path / "dit.safetensors"))
    pool = loader.ModelPool()
    with pytest.raises(ValueError, match="Cannot detect the model type"):
        pool.auto_load_model(path)
    loader.register_model_config({"model_hash": loader.hash_model_file(path), "model_name": "wan_video_dit",
                                  "model_class": "fairygen_amd.wan_video_dit.WanModel", "extra_kwargs": cfg})
    pool.auto_load_model(path)
    m = pool.fetch_model("wan_video_dit", index=2)
    assert isinstance(m, WanModel) and m.blocks[1].ffn[2].weight.dtype == torch.bfloat16
    assert torch.equal(m.state_dict()["blocks.0.modulation"], sd["blocks.0.modulation"])
    assert m.freqs[0].dtype == torch.complex128 and not m.freqs[0].is_meta
    assert pool.fetch_model("wan_video_vae") is None
    with pytest.raises(RuntimeError, match="offline"):
        ModelConfig(model_id="Wan-AI/Wan2.2-TI2V-5B", origin_file_pattern="Wan2.2_VAE.pth").download_if_necessary()
    with pytest.raises(ValueError):
        ModelConfig().download_if_necessary()


def test_scheduler_mirror(golden):
    g = golden("scheduler.safetensors")
    for n in (4, 30, 50):
        s = FlowMatchScheduler("Wan")
        s.set_timesteps(n, denoising_strength=1.0, shift=5.0)
        assert torch.equal(s.sigmas, g[f"sigmas_{n}"]) and torch.equal(s.timesteps, g[f"timesteps_{n}"])
    s = FlowMatchScheduler("Wan")
    s.set_timesteps(4, shift=5.0)
    sig, nxt = s.step_scalars(s.timesteps[3])
    assert float(nxt) == 0.0 and float(sig) == float(s.sigmas[3])
    sig, nxt = s.step_scalars(s.timesteps[1].to(torch.bfloat16))       # bf16-rounded timestep still finds its slot
    assert float(sig) == float(s.sigmas[1]) and float(nxt) == float(s.sigmas[2])
    with pytest.raises(NotImplementedError):
        FlowMatchScheduler("FLUX.1")


def test_lora_loader_mirror(golden):
    g = golden("lora.safetensors")
    cfg = synthetic.TINY_DIT_KWARGS
    shapes = synthetic.dit_shapes(cfg)
    m = WanModel(**cfg).to(torch.bfloat16)
    m.load_state_dict(synthetic.random_state_dict(shapes, seed=1234))
    lora = synthetic.random_lora(shapes, rank=4, seed=4321)
    ld = GeneralLoRALoader(device="cpu", torch_dtype=torch.bfloat16)
    conv = ld.convert_state_dict(lora)
    assert ld.convert_state_dict(conv).keys() == conv.keys()                      # idempotent
    up_down = {k.replace("lora_B.default", "lora_up.x").replace("lora_A.default", "lora_down.x"): v for k, v in lora.items()}
    assert set(ld.convert_state_dict({"diffusion_model." + k: v for k, v in up_down.items()})) == set(conv)
    m.blocks[0].fused_weights()
    assert ld.fuse_lora_to_base_model(m, conv, alpha=0.5) == 20
    assert m.blocks[0]._fused is None                                             # fused QKV cache invalidated
    fused = m.state_dict()
    for k in ("blocks.0.self_attn.q.weight", "blocks.1.cross_attn.v.weight", "blocks.1.ffn.2.weight", "blocks.0.ffn.0.weight",
              "blocks.0.self_attn.q.bias"):
        assert torch.equal(fused[k], g[k]), k
    stage2 = {k.replace(".lora_B.default.weight", ".lora_B2.weight"): seeded(v.shape, 77 + i, scale=0.02)
              for i, (k, v) in enumerate(sorted(lora.items())) if ".lora_B." in k}
    merged = merge_lora_weights(lora, stage2)
    assert torch.equal(merged["blocks.0.self_attn.q.lora_B.default.weight"], g["merged.blocks.0.self_attn.q.lora_B"])


def _cpu_pipe():
    pipe = WanVideoPipeline(device="cpu", torch_dtype=torch.bfloat16)
    pipe.dit = WanModel(**synthetic.TINY_DIT_KWARGS).to(torch.bfloat16)
    pipe.vae = WanVideoVAE38(dim=32, dec_dim=32).to(torch.bfloat16)
    pipe.height_division_factor = pipe.width_division_factor = pipe.vae.upsampling_factor * 2
    return pipe


def test_pipeline_units_and_call_surface(golden):
    pipe = _cpu_pipe()
    assert pipe.check_resize_height_width(470, 830, 80) == (480, 832, 81)
    assert torch.equal(pipe.generate_noise((1, 48, 2, 4, 4), seed=1), golden("pixels.safetensors")["noise_seed1"])
    shared = {"height": 64, "width": 64, "num_frames": 9, "seed": 1, "rand_device": "cpu", "cfg_scale": 5.0, "input_video": None,
              "input_image": None, "first_frame_latents": seeded((1, 48, 1, 4, 4), 4), "tiled": True, "tile_size": (3, 3),
              "tile_stride": (2, 2)}
    posi, nega = {"prompt": seeded((1, 16, 128), 2)}, {"negative_prompt": seeded((1, 16, 128), 3)}
    for unit in pipe.units:
        shared, posi, nega = pipe.unit_runner(unit, pipe, shared, posi, nega)
    assert shared["latents"].shape == (1, 48, 3, 4, 4) and shared["fuse_vae_embedding_in_latents"] is True
    assert torch.equal(shared["latents"][:, :, 0:1], shared["first_frame_latents"])
    assert torch.equal(posi["context"], seeded((1, 16, 128), 2)) and torch.equal(nega["context"], seeded((1, 16, 128), 3))
    # model_fn / vae.decode are the reference's plug points: swap them for recorders, the loop math is GPU-only
    with pytest.raises(RuntimeError, match="text encoder"):
        pipe(prompt="a pig walks", seed=1, height=64, width=64, num_frames=9)
    with pytest.raises(NotImplementedError):
        pipe(prompt=posi["prompt"], vace_video=[1], height=64, width=64, num_frames=9)
    with pytest.raises(TypeError):
        pipe(prompt=posi["prompt"], bogus=1)
    with pytest.raises(hip.HipLibraryError):        # the hot loop refuses to run on CPU tensors
        pipe(prompt=posi["prompt"], negative_prompt=nega["negative_prompt"], first_frame_latents=shared["first_frame_latents"],
             seed=1, height=64, width=64, num_frames=9, num_inference_steps=1, progress_bar_cmd=lambda x: x)


def test_tile_grid_matches_oracle():
    for hw in ((44, 80), (30, 52), (6, 8), (31, 53)):
        for size, stride in (((30, 52), (15, 26)), ((34, 34), (18, 16)), ((3, 4), (2, 2))):
            assert WanVideoVAE38.tile_tasks(*hw, size, stride) == wan_vae.tile_tasks(*hw, size, stride)
    assert len(WanVideoVAE38.tile_tasks(44, 80, (30, 52), (15, 26))) == 6
    assert len(WanVideoVAE38.tile_tasks(30, 52, (30, 52), (15, 26))) == 1


def test_teacache_rows_form_matches_dense():
    """The product keeps the time modulation as 2 distinct rows; its relative-L1 statistic must equal the reference's
    dense (1, N, 6, dim) computation, and the TeaCache state machine must take the same decisions from either form."""
    from fairygen_amd.wan_video import TeaCache, TimeModulation
    n, first, dim = 48, 16, 32
    dense_of = lambda rows: torch.cat([rows[0:1].expand(first, -1, -1), rows[1:2].expand(n - first, -1, -1)]).unsqueeze(0)  # noqa: E731
    seq = [seeded((2, 6, dim), 60 + i) * (1.0 + 0.05 * i) for i in range(6)]
    for a, b in zip(seq, seq[1:]):
        want = ((dense_of(b) - dense_of(a)).abs().mean() / dense_of(a).abs().mean()).item()
        assert TimeModulation(b, first, n).rel_l1_to(TimeModulation(a, first, n)) == want
    one = [seeded((1, 6, dim), 70 + i) for i in range(3)]
    want = ((one[1] - one[0]).abs().mean() / one[0].abs().mean()).item()
    assert TimeModulation(one[1], 0, n).rel_l1_to(TimeModulation(one[0], 0, n)) == want
    x = seeded((1, n, dim), 80)
    base = seeded((2, 6, dim), 90).float()
    seq = [(base * (1.0 + 0.02 * i)).to(torch.bfloat16) for i in range(6)]       # ~2 % change per step: rescaled ~0.16
    decisions = []
    for form in ("rows", "dense"):
        tc = TeaCache(len(seq), rel_l1_thresh=0.4, model_id="Wan2.1-I2V-14B-720P")
        got = []
        for rows in seq:
            skip = tc.check(None, x, TimeModulation(rows, first, n) if form == "rows" else dense_of(rows))
            got.append(skip)
            if not skip:
                tc.previous_residual = x             # what store() would leave behind
                tc.previous_hidden_states = None
        assert got[0] is False and got[-1] is False and tc.step == 0
        decisions.append(got)
    assert decisions[0] == decisions[1] == [False, True, True, False, True, False]
    with pytest.raises(ValueError, match="not a supported TeaCache model id"):
        TeaCache(4, 0.1, "Wan2.2-TI2V-5B")


def _mp4_boxes(data, off, end, path=()):
    """Flat {path: payload} map of an ISO base media file's box tree (containers descended)."""
    import struct
    out = {}
    while off < end:
        size, tag = struct.unpack(">I4s", data[off: off + 8])
        assert size >= 8 and off + size <= end
        here = path + (tag.decode("latin1"),)
        if tag in (b"moov", b"trak", b"mdia", b"minf", b"stbl", b"dinf"):
            out.update(_mp4_boxes(data, off + 8, off + size, here))
        else:
            out["/".join(here)] = data[off + 8: off + size]
        off += size
    return out


def test_save_video_fallback_honours_the_requested_file_name(tmp_path):
    """Without imageio/ffmpeg (this image) save_video writes Motion-JPEG into the container the file name asks for, AT that
    path: 'clip.mp4' -> an ISO base media file whose sample tables parse back (count, size, rate, chunk offset) and whose
    samples decode to the input within JPEG error; 'clip.avi' -> a RIFF AVI with header, index and frames."""
    import io
    import struct
    import numpy as np
    from PIL import Image
    from fairygen_amd import save_video
    rng = np.random.default_rng(0)
    base = rng.integers(0, 256, size=(8, 12, 3), dtype=np.uint8).repeat(8, 0).repeat(8, 1)        # 64x96 blocks
    frames = [Image.fromarray(np.roll(base, 8 * i, axis=1)) for i in range(5)]
    target = str(tmp_path / "clip.mp4")
    out = save_video(frames, target, fps=15, quality=9)
    assert out == target
    data = open(out, "rb").read()
    assert data[4:8] == b"ftyp" and data[8:12] == b"isom"
    boxes = _mp4_boxes(data, 0, len(data))
    stbl = "moov/trak/mdia/minf/stbl/"
    timescale, duration = struct.unpack(">II", boxes["moov/mvhd"][12:20])
    assert timescale == 15000 and duration == 5000
    w, h = struct.unpack(">II", boxes["moov/trak/tkhd"][-8:])
    assert (w >> 16, h >> 16) == (96, 64)
    assert boxes["moov/trak/mdia/hdlr"][8:12] == b"vide"
    stsd = boxes[stbl + "stsd"]
    assert stsd[12:16] == b"mp4v" and struct.unpack(">HH", stsd[16 + 24: 16 + 28]) == (96, 64) and b"esds" in stsd
    assert stsd[stsd.index(b"esds") + 8 + 5 + 2] == 0x6C       # ES_Descriptor(tag,len,id:2,flags) -> DecoderConfig(tag,len) -> OTI = JPEG
    assert struct.unpack(">III", boxes[stbl + "stts"][4:16]) == (1, 5, 1000)
    assert struct.unpack(">IIII", boxes[stbl + "stsc"][4:20]) == (1, 1, 5, 1)
    uniform, count = struct.unpack(">II", boxes[stbl + "stsz"][4:12])
    sizes = struct.unpack(">5I", boxes[stbl + "stsz"][12:32])
    (n_chunks, offset) = struct.unpack(">II", boxes[stbl + "stco"][4:12])
    assert (uniform, count, n_chunks) == (0, 5, 1) and data[offset - 4: offset] == b"mdat" and sum(sizes) == len(boxes["mdat"])
    for i, size in enumerate(sizes):
        got = np.array(Image.open(io.BytesIO(data[offset: offset + size])).convert("RGB")).astype(int)
        assert np.abs(got - np.array(frames[i]).astype(int)).mean() < 12
        offset += size
    # .avi (and any other extension): RIFF AVI at exactly that path
    target = str(tmp_path / "clip.avi")
    out = save_video(frames, target, fps=15, quality=9)
    assert out == target
    data = open(out, "rb").read()
    assert data[:4] == b"RIFF" and data[8:12] == b"AVI " and struct.unpack("<I", data[4:8])[0] == len(data) - 8
    avih = data.index(b"avih")
    usec, _, _, _, n, _, streams, _, w, h = struct.unpack("<10I", data[avih + 8: avih + 48])
    assert (n, streams, w, h) == (5, 1, 96, 64) and abs(1e6 / usec - 15) < 0.01
    movi = data.index(b"movi")
    idx1 = data.index(b"idx1", movi)
    entries = struct.unpack("<I", data[idx1 + 4: idx1 + 8])[0] // 16
    assert entries == 5
    for i in range(entries):
        tag, _, off, size = struct.unpack("<4sIII", data[idx1 + 8 + 16 * i: idx1 + 24 + 16 * i])
        start = movi + off                                              # offsets are relative to the 'movi' fourcc
        assert tag == b"00dc" and data[start: start + 4] == b"00dc" and struct.unpack("<I", data[start + 4: start + 8])[0] == size
        got = np.array(Image.open(io.BytesIO(data[start + 8: start + 8 + size])).convert("RGB")).astype(int)
        assert np.abs(got - np.array(frames[i]).astype(int)).mean() < 12
    with pytest.raises(ValueError):
        save_video([], str(tmp_path / "empty.mp4"), fps=15)

"""WanVideoPipeline for the FairyGen animation hot path on MI355X.

Drop-in for what ``animation/inference.py:9-30`` touches of ``diffsynth/pipelines/wan_video.py``:
``WanVideoPipeline.from_pretrained(torch_dtype, device, model_configs=[ModelConfig...])`` :98-169,
``pipe.load_lora(pipe.dit, path, alpha=)`` (``diffusion/base_pipeline.py:231-266``),
``pipe(prompt=, negative_prompt=, input_image=, num_frames=, seed=, tiled=, ...)`` :173-329 and the plug points
``pipe.model_fn`` / ``pipe.units`` / ``pipe.dit`` / ``pipe.vae`` / ``pipe.scheduler``.
``model_fn_wan_video`` is the TI2V-5B branch of :1122-1388; the other Wan variants' units (S2V, VACE, VAP,
Animate, FunControl, camera, LongCat) are out of scope and their kwargs raise.

Extensions (the only additions to the call surface, both optional):
  * ``prompt`` / ``negative_prompt`` may be pre-embedded context tensors ``(1, L, text_dim)`` (bench / tests: no
    tokenizer assets offline); strings go through ``pipe.tokenizer`` + the umT5 encoder like in the reference;
  * ``first_frame_latents=`` supplies the TI2V conditioning latent directly (bench / tests); ``input_image=`` goes
    through ``pipe.vae.encode`` (VAE38 encoder on the same HIP kernels) exactly like the reference.
"""
import numpy as np
import os

import torch
import torch.nn.functional as F
from PIL import Image
from tqdm import tqdm

from . import hip
from .flow_match import FlowMatchScheduler
from .loader import ModelConfig, ModelPool, load_state_dict
from .lora import GeneralLoRALoader
from .wan_video_dit import WanModel, sinusoidal_embedding_1d


# ------------------------------------------------------------------------------------ TeaCache
class TimeModulation:
    """The time modulation t_mod in the form this build keeps it: `rows` (R, 6, dim) distinct rows instead of the
    reference's per-token (1, N, 6, dim) tensor — R = 2 in TI2V mode (row 0 for the `first_rows` tokens of the pinned
    first latent frame, row 1 for the other n - first_rows), R = 1 otherwise."""

    def __init__(self, rows, first_rows, n):
        self.rows, self.first_rows, self.n = rows, int(first_rows), int(n)

    def clone(self):
        return TimeModulation(self.rows.clone(), self.first_rows, self.n)

    def rel_l1_to(self, prev):
        """((self - prev).abs().mean() / prev.abs().mean()).item() of the dense tensors, computed from the rows: the bf16
        elementwise difference, fp32 sums weighted by how many tokens use each row, bf16 means, bf16 quotient."""
        d = (self.rows - prev.rows).abs().float().sum(dim=(1, 2))
        p = prev.rows.abs().float().sum(dim=(1, 2))
        if self.rows.shape[0] == 2:
            w = torch.tensor([self.first_rows, self.n - self.first_rows], dtype=torch.float32, device=d.device)
            d, p, count = (d * w).sum(), (p * w).sum(), float(self.n)
        else:
            d, p, count = d.sum(), p.sum(), 1.0
        per = count * self.rows.shape[1] * self.rows.shape[2]
        dt = self.rows.dtype
        return ((d / per).to(dt) / (p / per).to(dt)).cpu().item()


class TeaCache:
    """pipelines/wan_video.py:1016-1065: skip the DiT blocks of a step while the accumulated polynomial-rescaled
    relative L1 change of the time modulation stays under `rel_l1_thresh`; one instance per CFG branch."""

    coefficients_dict = {
        "Wan2.1-T2V-1.3B": [-5.21862437e+04, 9.23041404e+03, -5.28275948e+02, 1.36987616e+01, -4.99875664e-02],
        "Wan2.1-T2V-14B": [-3.03318725e+05, 4.90537029e+04, -2.65530556e+03, 5.87365115e+01, -3.15583525e-01],
        "Wan2.1-I2V-14B-480P": [2.57151496e+05, -3.54229917e+04, 1.40286849e+03, -1.35890334e+01, 1.32517977e-01],
        "Wan2.1-I2V-14B-720P": [8.10705460e+03, 2.13393892e+03, -3.72934672e+02, 1.66203073e+01, -4.17769401e-02],
    }

    def __init__(self, num_inference_steps, rel_l1_thresh, model_id):
        if model_id not in self.coefficients_dict:
            supported = ", ".join(self.coefficients_dict)
            raise ValueError(f"{model_id} is not a supported TeaCache model id. Please choose a valid model id in ({supported}).")
        self.num_inference_steps, self.rel_l1_thresh = num_inference_steps, rel_l1_thresh
        self.coefficients = self.coefficients_dict[model_id]
        self.step = 0
        self.accumulated_rel_l1_distance = 0
        self.previous_modulated_input = self.previous_residual = self.previous_hidden_states = None

    def check(self, dit, x, t_mod):
        """True -> skip the blocks this step.  t_mod: TimeModulation (or a dense tensor laid out like the reference's)."""
        modulated_inp = t_mod.clone()
        if self.step == 0 or self.step == self.num_inference_steps - 1:
            should_calc = True
            self.accumulated_rel_l1_distance = 0
        else:
            if isinstance(modulated_inp, TimeModulation):
                rel = modulated_inp.rel_l1_to(self.previous_modulated_input)
            else:
                prev = self.previous_modulated_input
                rel = ((modulated_inp - prev).abs().mean() / prev.abs().mean()).cpu().item()
            self.accumulated_rel_l1_distance += np.poly1d(self.coefficients)(rel)
            should_calc = not (self.accumulated_rel_l1_distance < self.rel_l1_thresh)
            if should_calc:
                self.accumulated_rel_l1_distance = 0
        self.previous_modulated_input = modulated_inp
        self.step += 1
        if self.step == self.num_inference_steps:
            self.step = 0
        if should_calc:
            self.previous_hidden_states = x.clone()
        return not should_calc

    def store(self, hidden_states):
        self.previous_residual = hidden_states - self.previous_hidden_states
        self.previous_hidden_states = None

    def update(self, hidden_states):
        return hip.gate_residual(hidden_states.contiguous(), self.previous_residual)      # x + residual on the device


# ------------------------------------------------------------------------------------ pipeline units
class PipelineUnit:
    """diffusion/base_pipeline.py:12-56 — declarative (inputs -> outputs) step run before the loop."""

    def __init__(self, seperate_cfg=False, take_over=False, input_params=None, output_params=None,
                 input_params_posi=None, input_params_nega=None, onload_model_names=None):
        self.seperate_cfg, self.take_over = seperate_cfg, take_over
        self.input_params, self.output_params = input_params, output_params
        self.input_params_posi, self.input_params_nega = input_params_posi, input_params_nega
        self.onload_model_names = onload_model_names

    def process(self, pipe, **kwargs):
        return {}


class PipelineUnitRunner:
    """diffusion/base_pipeline.py:411-441."""

    def __call__(self, unit, pipe, inputs_shared, inputs_posi, inputs_nega):
        if unit.take_over:
            return unit.process(pipe, inputs_shared=inputs_shared, inputs_posi=inputs_posi, inputs_nega=inputs_nega)
        if unit.seperate_cfg:
            shared = {n: inputs_shared.get(n) for n in (unit.input_params or ())}
            out = unit.process(pipe, **{n: inputs_posi.get(src) for n, src in unit.input_params_posi.items()}, **shared)
            inputs_posi.update(out)
            if inputs_shared["cfg_scale"] != 1:
                out = unit.process(pipe, **{n: inputs_nega.get(src) for n, src in unit.input_params_nega.items()}, **shared)
            inputs_nega.update(out)
        else:
            inputs_shared.update(unit.process(pipe, **{n: inputs_shared.get(n) for n in unit.input_params}))
        return inputs_shared, inputs_posi, inputs_nega


class WanVideoUnit_ShapeChecker(PipelineUnit):
    def __init__(self):
        super().__init__(input_params=("height", "width", "num_frames"), output_params=("height", "width", "num_frames"))

    def process(self, pipe, height, width, num_frames):
        height, width, num_frames = pipe.check_resize_height_width(height, width, num_frames)
        return {"height": height, "width": width, "num_frames": num_frames}


class WanVideoUnit_NoiseInitializer(PipelineUnit):
    def __init__(self):
        super().__init__(input_params=("height", "width", "num_frames", "seed", "rand_device"), output_params=("noise",))

    def process(self, pipe, height, width, num_frames, seed, rand_device):
        length = (num_frames - 1) // 4 + 1
        shape = (1, pipe.vae.model.z_dim, length, height // pipe.vae.upsampling_factor, width // pipe.vae.upsampling_factor)
        return {"noise": pipe.generate_noise(shape, seed=seed, rand_device=rand_device)}


class WanVideoUnit_PromptEmbedder(PipelineUnit):
    def __init__(self):
        super().__init__(seperate_cfg=True, input_params_posi={"prompt": "prompt"},
                         input_params_nega={"prompt": "negative_prompt"}, output_params=("context",),
                         onload_model_names=("text_encoder",))

    def process(self, pipe, prompt):
        if isinstance(prompt, torch.Tensor):          # pre-embedded context (extension, see module docstring)
            return {"context": prompt.to(dtype=pipe.torch_dtype, device=pipe.device)}
        if pipe.text_encoder is None or pipe.tokenizer is None:
            raise RuntimeError("no text encoder / tokenizer loaded (pass the umT5 checkpoint in model_configs and "
                               "tokenizer_config=ModelConfig(path=<local google/umt5-xxl dir>)), or pass prompt= and "
                               "negative_prompt= as pre-embedded (1, L, text_dim) context tensors")
        ids, mask = pipe.tokenizer(prompt, return_mask=True, add_special_tokens=True)
        ids, mask = ids.to(pipe.device), mask.to(pipe.device)
        seq_lens = mask.gt(0).sum(dim=1).long()
        emb = pipe.text_encoder(ids, mask)
        for v in seq_lens:
            emb[:, v:] = 0
        return {"context": emb}


class WanVideoUnit_InputVideoEmbedder(PipelineUnit):
    """:366-390 — text-to-video starts from the noise; video-to-video (`input_video=` frames + `denoising_strength`)
    from the VAE38-encoded video noised to the first timestep of the shortened schedule."""

    def __init__(self):
        super().__init__(input_params=("input_video", "noise", "tiled", "tile_size", "tile_stride"),
                         output_params=("latents", "input_latents"), onload_model_names=("vae",))

    def process(self, pipe, input_video, noise, tiled, tile_size, tile_stride):
        if input_video is None:
            return {"latents": noise}
        video = pipe.preprocess_video(input_video)
        input_latents = pipe.vae.encode(video, device=pipe.device, tiled=tiled, tile_size=tile_size,
                                        tile_stride=tile_stride).to(dtype=pipe.torch_dtype, device=pipe.device)
        return {"latents": pipe.scheduler.add_noise(input_latents, noise, timestep=pipe.scheduler.timesteps[0])}


class WanVideoUnit_ImageEmbedderFused(PipelineUnit):
    """TI2V-5B conditioning: first latent frame = encoded image, re-pinned every step (:479-497)."""

    def __init__(self):
        super().__init__(input_params=("input_image", "first_frame_latents", "latents", "height", "width", "tiled",
                                       "tile_size", "tile_stride"),
                         output_params=("latents", "fuse_vae_embedding_in_latents", "first_frame_latents"),
                         onload_model_names=("vae",))

    def process(self, pipe, input_image, first_frame_latents, latents, height, width, tiled, tile_size, tile_stride):
        if not pipe.dit.fuse_vae_embedding_in_latents or (input_image is None and first_frame_latents is None):
            return {}
        if first_frame_latents is None:
            image = pipe.preprocess_image(input_image.resize((width, height))).transpose(0, 1)
            first_frame_latents = pipe.vae.encode([image], device=pipe.device, tiled=tiled, tile_size=tile_size,
                                                  tile_stride=tile_stride)
        z = first_frame_latents.to(dtype=pipe.torch_dtype, device=pipe.device)
        latents[:, :, 0:1] = z
        return {"latents": latents, "fuse_vae_embedding_in_latents": True, "first_frame_latents": z}


class WanVideoUnit_TeaCache(PipelineUnit):
    """:769-781 — one TeaCache per CFG branch when tea_cache_l1_thresh is given."""

    def __init__(self):
        super().__init__(seperate_cfg=True,
                         input_params_posi={"num_inference_steps": "num_inference_steps", "tea_cache_l1_thresh": "tea_cache_l1_thresh",
                                            "tea_cache_model_id": "tea_cache_model_id"},
                         input_params_nega={"num_inference_steps": "num_inference_steps", "tea_cache_l1_thresh": "tea_cache_l1_thresh",
                                            "tea_cache_model_id": "tea_cache_model_id"},
                         output_params=("tea_cache",))

    def process(self, pipe, num_inference_steps, tea_cache_l1_thresh, tea_cache_model_id):
        if tea_cache_l1_thresh is None:
            return {}
        return {"tea_cache": TeaCache(num_inference_steps, rel_l1_thresh=tea_cache_l1_thresh, model_id=tea_cache_model_id)}


class WanVideoUnit_CfgMerger(PipelineUnit):
    """:785-803 — cfg_merge=True: the positive and the negative context are concatenated on the batch axis, the loop then
    makes ONE model_fn call per step and splits its (2, ...) prediction.  (Like the reference, the per-branch inputs —
    including a TeaCache — are dropped.)"""

    def __init__(self):
        super().__init__(take_over=True)
        self.concat_tensor_names = ["context"]

    def process(self, pipe, inputs_shared, inputs_posi, inputs_nega):
        if inputs_shared.get("cfg_merge"):
            # a per-branch tensor becomes a (2, ...) batch [positive, negative]; one that is already shared is doubled so that it lines up
            # with the batched ones; everything else the branches carried is dropped (the loop makes one call without branch inputs)
            merged = {}
            for name in self.concat_tensor_names:
                pair = (inputs_posi.get(name), inputs_nega.get(name))
                if pair[0] is None or pair[1] is None:
                    pair = (inputs_shared.get(name),) * 2
                if pair[0] is not None:
                    merged[name] = torch.cat(pair, dim=0)
            inputs_shared.update(merged)
            inputs_posi.clear()
            inputs_nega.clear()
        return inputs_shared, inputs_posi, inputs_nega


# parameters of other Wan variants' features: inert without the feature's main input (which raises below), so the
# reference's own non-None defaults — or any other value — are accepted
_INERT_KWARGS = ("audio_sample_rate", "camera_control_speed", "camera_control_origin", "vace_scale", "vap_prompt",
                 "negative_vap_prompt")
_OUT_OF_SCOPE_KWARGS = (
    "end_image", "input_audio", "audio_embeds", "s2v_pose_video", "s2v_pose_latents", "motion_video",
    "control_video", "reference_image", "camera_control_direction", "vace_video", "vace_video_mask",
    "vace_reference_image", "animate_pose_video", "animate_face_video", "animate_inpaint_video", "animate_mask_video",
    "vap_video", "motion_bucket_id", "longcat_video",
)


class WanVideoPipeline(torch.nn.Module):
    def __init__(self, device="cuda", torch_dtype=torch.bfloat16):
        super().__init__()
        self.device, self.torch_dtype = device, torch_dtype
        self.height_division_factor = self.width_division_factor = 16
        self.time_division_factor, self.time_division_remainder = 4, 1
        self.vram_management_enabled = False
        self.unit_runner = PipelineUnitRunner()
        self.lora_loader = GeneralLoRALoader
        self.scheduler = FlowMatchScheduler("Wan")
        self.tokenizer = None
        self.text_encoder = None
        self.dit: WanModel = None
        self.dit2 = None
        self.vae = None
        self.in_iteration_models = ("dit",)
        self.units = [
            WanVideoUnit_ShapeChecker(),
            WanVideoUnit_NoiseInitializer(),
            WanVideoUnit_PromptEmbedder(),
            WanVideoUnit_InputVideoEmbedder(),
            WanVideoUnit_ImageEmbedderFused(),
            WanVideoUnit_TeaCache(),
            WanVideoUnit_CfgMerger(),
        ]
        self.post_units = []
        self.model_fn = model_fn_wan_video
        self.sequence_shard = None          # set by enable_sequence_parallel(): token shard group of this rank
        self.parallel = None                # sequence_parallel.ParallelLayout (world = cfg_parallel x sp)
        self.use_unified_sequence_parallel = False

    # ----------------------------------------------------------------------------- construction
    @staticmethod
    def from_pretrained(torch_dtype=torch.bfloat16, device="cuda", model_configs=[], tokenizer_config=None,
                        audio_processor_config=None, redirect_common_files=True, use_usp=False, vram_limit=None):
        pipe = WanVideoPipeline(device=device, torch_dtype=torch_dtype)
        pool = pipe.download_and_load_models(model_configs, vram_limit)
        pipe.text_encoder = pool.fetch_model("wan_video_text_encoder")
        dit = pool.fetch_model("wan_video_dit", index=2)
        if isinstance(dit, list):
            raise NotImplementedError("two-expert (dit/dit2) Wan2.2-A14B checkpoints are outside the TI2V-5B hot path")
        pipe.dit = dit
        pipe.vae = pool.fetch_model("wan_video_vae")
        if pipe.vae is not None:
            pipe.height_division_factor = pipe.width_division_factor = pipe.vae.upsampling_factor * 2
        if tokenizer_config is not None:      # reference default downloads google/umt5-xxl; here: ModelConfig(path=<local dir>)
            tokenizer_config.download_if_necessary()
            from .wan_video_text_encoder import HuggingfaceTokenizer
            pipe.tokenizer = HuggingfaceTokenizer(name=tokenizer_config.path, seq_len=512, clean="whitespace")
        if use_usp:
            pipe.enable_usp()
        return pipe

    def download_and_load_models(self, model_configs=[], vram_limit=None):
        pool = ModelPool()
        for cfg in model_configs:
            cfg.download_if_necessary()
            vram = cfg.vram_config()
            vram["computation_dtype"] = vram["computation_dtype"] or self.torch_dtype
            vram["computation_device"] = vram["computation_device"] or self.device
            pool.auto_load_model(cfg.path, vram_config=vram, vram_limit=vram_limit, clear_parameters=cfg.clear_parameters)
        return pool

    def enable_usp(self):
        """The reference's use_usp=True monkey-patches xfuser Ulysses attention (:84-95); here the same switch
        turns on latent-temporal token sharding over RCCL with the same Ulysses exchange (sequence_parallel.py)."""
        return self.enable_sequence_parallel(attn_mode="ulysses" if self.dit is None or self._ulysses_ok() else "allgather")

    def _ulysses_ok(self):
        import torch.distributed as dist
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        return self.dit.num_heads % world == 0

    def enable_sequence_parallel(self, cfg_parallel=1, attn_mode="allgather", layout=None):
        """Shard the denoise loop over the default process group: world = cfg_parallel x sp.  cfg_parallel = 2 gives
        each half of the ranks one CFG branch; inside a half (or the whole world) tokens are sharded by
        latent-temporal ranges and self-attention exchanges by `attn_mode` ("allgather" K/V or "ulysses").
        layout: a ready sequence_parallel.ParallelLayout (e.g. one replica's, fairygen_amd.batch) instead."""
        from .sequence_parallel import ParallelLayout
        self.parallel = layout if layout is not None else ParallelLayout(cfg_parallel, attn_mode)
        self.sequence_shard = self.parallel.shard
        self.use_unified_sequence_parallel = True
        return self

    # ----------------------------------------------------------------------------- BasePipeline helpers
    def check_resize_height_width(self, height, width, num_frames=None):
        hf, wf = self.height_division_factor, self.width_division_factor
        if height % hf != 0:
            height = (height + hf - 1) // hf * hf
            print(f"height % {hf} != 0. We round it up to {height}.")
        if width % wf != 0:
            width = (width + wf - 1) // wf * wf
            print(f"width % {wf} != 0. We round it up to {width}.")
        if num_frames is None:
            return height, width
        tf, tr = self.time_division_factor, self.time_division_remainder
        if num_frames % tf != tr:
            num_frames = (num_frames + tf - 1) // tf * tf + tr
            print(f"num_frames % {tf} != {tr}. We round it up to {num_frames}.")
        return height, width, num_frames

    def preprocess_image(self, image, torch_dtype=None, device=None, min_value=-1, max_value=1):
        """PIL -> (1,C,H,W) in [min,max] (base_pipeline.py:112-118)."""
        x = torch.Tensor(np.array(image, dtype=np.float32)).to(dtype=torch_dtype or self.torch_dtype, device=device or self.device)
        x = x * ((max_value - min_value) / 255) + min_value
        return x.permute(2, 0, 1).unsqueeze(0)

    def preprocess_video(self, video, torch_dtype=None, device=None, pattern="B C T H W", min_value=-1, max_value=1):
        """list of PIL frames -> (1,C,T,H,W) in [min,max] (base_pipeline.py:121-125)."""
        frames = [self.preprocess_image(im, torch_dtype=torch_dtype, device=device, min_value=min_value, max_value=max_value)
                  for im in video]
        return torch.stack(frames, dim=pattern.index("T") // 2)

    def generate_noise(self, shape, seed=None, rand_device="cpu", rand_torch_dtype=torch.float32, device=None, torch_dtype=None):
        generator = None if seed is None else torch.Generator(rand_device).manual_seed(seed)
        noise = torch.randn(shape, generator=generator, device=rand_device, dtype=rand_torch_dtype)
        return noise.to(dtype=torch_dtype or self.torch_dtype, device=device or self.device)

    def vae_output_to_video(self, vae_output):
        """(1,3,F,H,W) in [-1,1] -> list of PIL frames; the uint8 conversion (truncation) runs on the device."""
        frames = hip.video_to_uint8(vae_output[0].contiguous()).cpu().numpy()
        return [Image.fromarray(f) for f in frames]

    def load_models_to_device(self, model_names):
        pass        # models stay resident in HBM (no VRAM management on this path)

    def load_lora(self, module, lora_config=None, alpha=1, hotload=None, state_dict=None):
        """base_pipeline.py:231-266.  hotload=None / False: fuse into the base weights (what inference.py does).
        hotload=True: keep the adapter unfused next to the Linear (AutoWrappedLinear.lora_forward semantics,
        core/vram/layers.py:417-436) so `clear_lora()` can drop it again.  The reference only allows that on modules
        under its VRAM manager; here the weights are always resident in HBM, so any module with `add_hot_lora` takes it."""
        if state_dict is None:
            if isinstance(lora_config, str):
                lora = load_state_dict(lora_config, torch_dtype=self.torch_dtype, device=self.device)
            else:
                lora_config.download_if_necessary()
                lora = load_state_dict(lora_config.path, torch_dtype=self.torch_dtype, device=self.device)
        else:
            lora = state_dict
        loader = self.lora_loader(torch_dtype=self.torch_dtype, device=self.device)
        lora = loader.convert_state_dict(lora)
        if hotload:
            if not hasattr(module, "add_hot_lora"):
                raise ValueError("VRAM Management is not enabled. LoRA hotloading is not supported.")
            updated = 0
            for name, sub in module.named_modules():
                a_key, b_key = f"{name}.lora_A.weight", f"{name}.lora_B.weight"
                if isinstance(sub, torch.nn.Linear) and a_key in lora and b_key in lora:
                    module.add_hot_lora(name, lora[a_key].to(self.torch_dtype) * alpha, lora[b_key])
                    updated += 1
            print(f"{updated} tensors are patched by LoRA. You can use `pipe.clear_lora()` to clear all LoRA layers.")
        else:
            loader.fuse_lora_to_base_model(module, lora, alpha=alpha)

    def clear_lora(self):
        """base_pipeline.py:269-279: drop every hot-loaded adapter (fused ones cannot be cleared, as in the reference)."""
        cleared = sum(m.clear_hot_loras() for m in self.modules() if hasattr(m, "clear_hot_loras"))
        print(f"{cleared} LoRA layers are cleared.")

    # ----------------------------------------------------------------------------- __call__
    @torch.no_grad()
    def __call__(self, prompt, negative_prompt="", input_image=None, input_video=None, seed=None, rand_device="cpu", height=480, width=832,
                 num_frames=81, cfg_scale=5.0, cfg_merge=False, switch_DiT_boundary=0.875, num_inference_steps=50,
                 sigma_shift=5.0, denoising_strength=1.0, tiled=True, tile_size=(30, 52), tile_stride=(15, 26),
                 sliding_window_size=None, sliding_window_stride=None, tea_cache_l1_thresh=None, tea_cache_model_id="",
                 progress_bar_cmd=tqdm, output_type="quantized",
                 first_frame_latents=None, **other):
        for name, value in other.items():
            if name in _INERT_KWARGS:
                continue
            if name not in _OUT_OF_SCOPE_KWARGS:
                raise TypeError(f"__call__() got an unexpected keyword argument {name!r}")
            if value is not None:
                raise NotImplementedError(f"{name}= belongs to another Wan variant / feature outside the TI2V-5B hot path")
        self.scheduler.set_timesteps(num_inference_steps, denoising_strength=denoising_strength, shift=sigma_shift)
        if seed is None and self.parallel is not None and self.parallel.world is not None and self.parallel.world.active:
            # the reference default (seed=None) would give every rank of the clip its own noise and the token shards
            # would silently mix different clips: the ranks of a layout agree on rank 0's draw
            seed = self.parallel.world.shared_seed()

        tea = {"tea_cache_l1_thresh": tea_cache_l1_thresh, "tea_cache_model_id": tea_cache_model_id,
               "num_inference_steps": num_inference_steps}
        inputs_posi = {"prompt": prompt, **tea}
        inputs_nega = {"negative_prompt": negative_prompt, **tea}
        inputs_shared = {
            "input_image": input_image, "first_frame_latents": first_frame_latents, "input_video": input_video,
            "seed": seed, "rand_device": rand_device, "height": height, "width": width, "num_frames": num_frames,
            "cfg_scale": cfg_scale, "cfg_merge": cfg_merge, "sigma_shift": sigma_shift,
            "tiled": tiled, "tile_size": tile_size, "tile_stride": tile_stride,
            "sliding_window_size": sliding_window_size, "sliding_window_stride": sliding_window_stride,
        }
        for unit in self.units:
            inputs_shared, inputs_posi, inputs_nega = self.unit_runner(unit, self, inputs_shared, inputs_posi, inputs_nega)

        # Denoise (reference :283-309)
        latents = self.denoise(inputs_shared, inputs_posi, inputs_nega, cfg_scale, progress_bar_cmd)
        inputs_shared["latents"] = latents
        for unit in self.post_units:
            inputs_shared, _, _ = self.unit_runner(unit, self, inputs_shared, inputs_posi, inputs_nega)

        # Decode (reference :322-325)
        video = self.decode_latents(inputs_shared["latents"], tiled=tiled, tile_size=tile_size, tile_stride=tile_stride)
        if output_type == "quantized":
            video = self.vae_output_to_video(video)
        return video

    def decode_latents(self, latents, tiled=True, tile_size=(30, 52), tile_stride=(15, 26)):
        """vae.decode on the device (reference :322-323); tiles are dealt over all ranks."""
        shard = self.parallel.world if (self.parallel is not None and self.parallel.world.active) else None
        return self.vae.decode(latents, device=self.device, tiled=tiled, tile_size=tile_size, tile_stride=tile_stride,
                               shard=shard)

    def denoise(self, inputs_shared, inputs_posi, inputs_nega, cfg_scale, progress_bar_cmd=tqdm):
        """The hot loop: per step forward(+), forward(-), then CFG combine + Euler step fused in one HIP kernel,
        then the first latent frame re-pinned (TI2V).  How the two forwards of a step are issued depends on the layout:

        * single GPU: one after the other;
        * cfg_parallel = 2: this rank's half of the world computes ONE branch, one world all-gather exchanges the predictions;
        * token-sharded, cfg_parallel = 1: the two branches — independent until the combine — are advanced in lockstep on
          ONE stream, each yielding right after it has started an exchange: branch B's GEMMs run while A's tensors cross
          xGMI, A's attention / FFN while B's do.  Collectives keep one program order on every rank."""
        models = {name: getattr(self, name) for name in self.in_iteration_models}
        shared = {k: v for k, v in inputs_shared.items()
                  if k in ("latents", "fuse_vae_embedding_in_latents", "sliding_window_size", "sliding_window_stride")}
        cfg_merge = bool(inputs_shared.get("cfg_merge")) and "context" not in inputs_posi
        if cfg_merge:
            shared["context"] = inputs_shared["context"]              # (2, L, text_dim): [positive; negative]
        shared["sequence_shard"] = self.sequence_shard
        latents = inputs_shared["latents"].contiguous()
        first = inputs_shared.get("first_frame_latents")
        sharded = self.sequence_shard is not None and self.sequence_shard.active
        cfg_split = self.parallel is not None and self.parallel.cfg_parallel == 2 and self.model_fn is model_fn_wan_video
        if cfg_merge and cfg_split:
            raise NotImplementedError("cfg_merge=True batches both branches into one forward; a cfg_parallel=2 layout gives "
                                      "each branch its own ranks — use one or the other")
        interleave = sharded and cfg_scale != 1.0 and self.model_fn is model_fn_wan_video and not cfg_split and not cfg_merge
        # the two CFG forwards of a step differ only in the context: what precedes block 0's cross-attention is computed once
        # (WanModel.forward_tokens_steps, cfg_prefix); not with TeaCache (per-branch state) or sliding windows (many forwards)
        share = (CFG_SHARE_PREFIX and cfg_scale != 1.0 and self.model_fn is model_fn_wan_video and not cfg_split and not cfg_merge
                 and inputs_posi.get("tea_cache") is None and inputs_nega.get("tea_cache") is None
                 and shared.get("sliding_window_size") is None)
        if CROSS_KV_CACHE and self.model_fn is model_fn_wan_video and shared.get("sliding_window_size") is None:
            shared["kv_cache"] = {}      # lives for this loop only: the prompt embeddings and the weights do not change inside it
        for progress_id, timestep in enumerate(progress_bar_cmd(self.scheduler.timesteps)):
            ts = timestep.unsqueeze(0).to(dtype=self.torch_dtype)       # bf16 rounding of t (:293), kept on the host
            shared["latents"] = latents
            if share:
                shared["cfg_prefix"] = {}
            if cfg_merge:
                # reference :296-299: one call on the batched context, then chunk (with cfg_scale == 1 the reference still
                # runs the batch of two and keeps both halves' mean-free "posi": it uses the whole (2, ...) tensor — not a
                # combination anyone relies on, so cfg_scale == 1 simply uses the positive half)
                both = self.model_fn(**models, **shared, timestep=ts)
                posi, nega = (t.contiguous() for t in both.chunk(2, dim=0))
                if cfg_scale == 1.0:
                    nega = None
            elif cfg_split:
                # this rank's half of the world computes ONE branch; one world all-gather exchanges the predictions
                mine = inputs_posi if (self.parallel.branch == 0 or cfg_scale == 1.0) else inputs_nega
                out_loc, grid = self.model_fn(**models, **shared, **inputs_posi_ctx(mine), timestep=ts, gather_output=False)
                both = self.parallel.gather_branches(out_loc, grid[0] * grid[1] * grid[2])
                posi = self.dit.unpatchify(both[0:1], grid).contiguous()
                nega = self.dit.unpatchify(both[1:2], grid).contiguous() if cfg_scale != 1.0 else None
            elif interleave:
                gens = [model_fn_wan_video_steps(**models, **shared, **inputs_posi_ctx(c), timestep=ts)
                        for c in (inputs_posi, inputs_nega)]
                posi, nega = (o.contiguous() for o in run_interleaved(gens))
            else:
                posi = self.model_fn(**models, **shared, **inputs_posi_ctx(inputs_posi), timestep=ts).contiguous()
                nega = self.model_fn(**models, **shared, **inputs_posi_ctx(inputs_nega), timestep=ts).contiguous() \
                    if cfg_scale != 1.0 else None
            sigma, sigma_next = self.scheduler.step_scalars(self.scheduler.timesteps[progress_id])
            latents = hip.cfg_euler(latents, posi, nega, cfg_scale, float(sigma_next - sigma))
            if first is not None:
                latents[:, :, 0:1] = first
        return latents


CFG_SHARE_PREFIX = os.environ.get("FAIRYGEN_CFG_SHARE", "1") != "0"
CROSS_KV_CACHE = os.environ.get("FAIRYGEN_CROSS_KV_CACHE", "1") != "0"


def inputs_posi_ctx(d):
    """The per-branch model_fn inputs: the embedded prompt and, when enabled, this branch's TeaCache."""
    out = {"context": d["context"]}
    if d.get("tea_cache") is not None:
        out["tea_cache"] = d["tea_cache"]
    return out


# ------------------------------------------------------------------------------------- the DiT forward
def model_fn_wan_video(dit, latents=None, timestep=None, context=None, fuse_vae_embedding_in_latents=False,
                       sequence_shard=None, gather_output=True, **kwargs):
    """One DiT forward (TI2V-5B / T2V branches of pipelines/wan_video.py:1217-1388); see model_fn_wan_video_steps."""
    gen = model_fn_wan_video_steps(dit, latents=latents, timestep=timestep, context=context,
                                   fuse_vae_embedding_in_latents=fuse_vae_embedding_in_latents,
                                   sequence_shard=sequence_shard, gather_output=gather_output, **kwargs)
    while True:
        try:
            next(gen)
        except StopIteration as done:
            return done.value


def run_interleaved(generators):
    """Advance several independent step-generators round-robin until all are exhausted; returns their values."""
    results, live = [None] * len(generators), list(range(len(generators)))
    while live:
        for i in list(live):
            try:
                next(generators[i])
            except StopIteration as done:
                results[i] = done.value
                live.remove(i)
    return results


def temporal_windows(t_all, sliding_window_size, sliding_window_stride):
    """The (t, t_) latent-frame ranges TemporalTiler_BCTHW.run visits (pipelines/wan_video.py:1097-1100)."""
    out = []
    for t in range(0, t_all, sliding_window_stride):
        if t - sliding_window_stride >= 0 and t - sliding_window_stride + sliding_window_size >= t_all:
            continue
        out.append((t, min(t + sliding_window_size, t_all)))
    return out


def temporal_tiler_steps(window_fn, latents, sliding_window_size, sliding_window_stride, window_shard=None):
    """TemporalTiler_BCTHW.run (pipelines/wan_video.py:1069-1118) as a generator: overlapping windows of latent frames,
    each an independent forward (`window_fn(window)` is a step generator), blended with linear ramps of width
    size - stride; accumulation in the data dtype on the device, in the reference's order.

    window_shard (a multi-rank sequence_parallel.TokenShard with attn_mode "windows"): the windows are dealt round-robin to
    the ranks — latent-temporal shards whose overlap frames are the halo — each rank runs its windows' forwards with no exchange
    inside them, one all-gather per call collects the window outputs, and every rank blends all of them in the reference's
    order: bit-identical to the single-GPU sliding-window result."""
    b, c, t_all, h, w = latents.shape
    value = torch.zeros((b, c, t_all, h, w), dtype=latents.dtype, device=latents.device)
    weight = torch.zeros((1, 1, t_all, 1, 1), dtype=latents.dtype, device=latents.device)
    border = sliding_window_size - sliding_window_stride
    windows = temporal_windows(t_all, sliding_window_size, sliding_window_stride)
    world, rank = (window_shard.world_size, window_shard.rank) if window_shard is not None else (1, 0)
    exchange = window_shard is not None and window_shard.active
    outs = {}
    for i, (t, t_) in enumerate(windows):
        if i % world == rank:
            outs[i] = yield from window_fn(latents[:, :, t:t_].contiguous())
    if exchange:
        # one collective: every rank contributes its windows, zero-padded to `sliding_window_size` frames, in slot order
        per_rank = (len(windows) + world - 1) // world
        slot = c * sliding_window_size * h * w
        send = torch.zeros((per_rank, slot), dtype=latents.dtype, device=latents.device)
        for i, o in outs.items():
            send[i // world, : o.numel()] = o.reshape(-1)
        full = window_shard._gather_rows(send, per_rank).view(world, per_rank, slot)
        for i, (t, t_) in enumerate(windows):
            outs[i] = full[i % world, i // world, : c * (t_ - t) * h * w].view(b, c, t_ - t, h, w)
    for i, (t, t_) in enumerate(windows):
        mask = torch.ones((t_ - t,))
        if border > 0:
            ramp = (torch.arange(border) + 0.5) / border
            if t != 0:
                mask[:border] = ramp
            if t_ != t_all:
                mask[-border:] = torch.flip(ramp, dims=(0,))
        mask = mask.view(1, 1, -1, 1, 1).to(device=latents.device, dtype=latents.dtype)
        value[:, :, t:t_] += outs[i] * mask
        weight[:, :, t:t_] += mask
    value /= weight
    return value


def model_fn_wan_video_steps(dit, latents=None, timestep=None, context=None, fuse_vae_embedding_in_latents=False,
                             sequence_shard=None, gather_output=True, tea_cache=None, sliding_window_size=None,
                             sliding_window_stride=None, cfg_prefix=None, kv_cache=None, **kwargs):
    """Generator form of the forward (yields where WanModel.forward_tokens_steps yields; returns the prediction, or
    with gather_output=False the head output of this rank's tokens (1, n_local, out_dim*prod(patch)) and the grid).

    timestep: (1,) tensor already rounded to the pipeline dtype (host or device).  The per-token time embedding of
    the reference (:1219-1228) has only two distinct rows (t=0 for the first latent frame, t elsewhere): both rows
    go through time_embedding / time_projection once and the kernels index them by token position.
    """
    assert latents.shape[0] == 1, "one clip per call"
    if context.shape[0] > 1:
        # merged CFG (reference :1237-1242): the latents are repeated over the context batch and the DiT runs on the batch.
        # Every op of the forward is per batch element, so the batch is evaluated element by element on the batch-1
        # kernels: identical results, the prediction comes back stacked on dim 0 like the reference's.
        if tea_cache is not None:
            raise NotImplementedError("TeaCache is per CFG branch; the merged call has none (the reference drops it too)")
        outs = []
        # the elements differ only in their context: block 0's self-attention half is shared — between exactly two (one producer, one consumer)
        prefix = {} if (CFG_SHARE_PREFIX and sliding_window_size is None and context.shape[0] == 2) else None
        for b in range(context.shape[0]):
            outs.append((yield from model_fn_wan_video_steps(
                dit, latents=latents, timestep=timestep, context=context[b:b + 1], sequence_shard=sequence_shard,
                fuse_vae_embedding_in_latents=fuse_vae_embedding_in_latents, gather_output=gather_output,
                sliding_window_size=sliding_window_size, sliding_window_stride=sliding_window_stride, cfg_prefix=prefix,
                kv_cache=kv_cache)))
        if not gather_output:
            return torch.cat([o[0] for o in outs], dim=0), outs[0][1]
        return torch.cat(outs, dim=0)
    if sliding_window_size is not None and sliding_window_stride is not None:
        # the reference's approximate long-video mode (:1158-1182).  Its window calls do not receive
        # fuse_vae_embedding_in_latents (absent from its model_kwargs), so they run in the single-timestep mode.
        if not gather_output:
            raise NotImplementedError("sliding windows need whole predictions: use a cfg_parallel=1 layout")
        by_window = sequence_shard is not None and sequence_shard.active and sequence_shard.attn_mode == "windows"
        if by_window and tea_cache is not None:
            raise NotImplementedError("TeaCache carries state from window to window (the reference shares one object across "
                                      "them); windows dealt to different ranks cannot reproduce that order")
        return (yield from temporal_tiler_steps(
            lambda win: model_fn_wan_video_steps(dit, latents=win, timestep=timestep, context=context,
                                                 sequence_shard=None if by_window else sequence_shard, tea_cache=tea_cache),
            latents, sliding_window_size, sliding_window_stride, sequence_shard if by_window else None))
    if sequence_shard is not None and sequence_shard.active and sequence_shard.attn_mode == "windows":
        raise ValueError('attn_mode="windows" shards the sliding-window mode: pass sliding_window_size= and '
                         'sliding_window_stride= (the exact path shards with "ulysses" or "allgather")')
    dev, dt = latents.device, latents.dtype
    tval = timestep.detach().to("cpu")
    ti2v = dit.seperated_timestep and fuse_vae_embedding_in_latents
    t_pos = torch.cat([torch.zeros(1, dtype=tval.dtype), tval]) if ti2v else tval
    emb = sinusoidal_embedding_1d(dit.freq_dim, t_pos).to(device=dev, dtype=dt)         # (R, freq_dim)
    te = dit.time_embedding
    t_rows = F.linear(hip.activation(F.linear(emb, te[0].weight, te[0].bias), "silu"), te[2].weight, te[2].bias)
    proj = dit.time_projection[1]
    mod_rows_t = F.linear(hip.activation(t_rows.clone(), "silu"), proj.weight, proj.bias).unflatten(1, (6, dit.dim))

    # what depends on the prompt embedding and the weights only — the text_embedding MLP here, the cross-attention K / V of every
    # block in forward_tokens_steps — is kept for the denoise loop that owns kv_cache (keyed by the prompt tensor)
    kv = None
    if kv_cache is not None and sliding_window_size is None:
        # an entry belongs to ONE prompt tensor object at ONE version (it keeps the tensor alive, so its address cannot be reused while the
        # entry exists); a loop has two prompts, so a handful of entries is a caller re-materialising its context every step: the oldest goes
        entries = kv_cache.setdefault("entries", [])
        base = context._base if context._base is not None else context          # the merged call passes views context[b:b+1] of one tensor
        ident = (context.storage_offset(), tuple(context.shape), tuple(context.stride()), context._version)
        kv = next((e["kv"] for e in entries if e["base"] is base and e["ident"] == ident), None)
        if kv is None:
            kv = {}
            entries.append({"base": base, "ident": ident, "kv": kv})
            del entries[:-4]
    if kv is not None and "ctx" in kv:
        ctx = kv["ctx"]
    else:
        tx = dit.text_embedding
        ctx = F.linear(hip.activation(F.linear(context, tx[0].weight, tx[0].bias), "gelu_tanh"), tx[2].weight, tx[2].bias)
        if kv is not None:
            kv["ctx"] = ctx

    x, (f, h, w) = dit.patchify(latents)
    n = f * h * w
    first_rows = h * w if ti2v else 0
    cos, sin = dit.rope_tables(f, h, w, dev)
    if sequence_shard is not None and sequence_shard.active:
        lo, hi = sequence_shard.local_range(n)
        x_loc = x[:, lo:hi].contiguous()
        skip = tea_cache is not None and tea_cache.check(dit, x_loc, TimeModulation(mod_rows_t, first_rows, n))
        out_loc = yield from dit.forward_tokens_steps(x_loc, ctx, mod_rows_t, t_rows, min(max(first_rows - lo, 0), hi - lo),
                                                      (cos[lo:hi].contiguous(), sin[lo:hi].contiguous() if sin is not None else None), sequence_shard, n,
                                                      tea_cache, skip, cfg_prefix if tea_cache is None else None, kv)
        if not gather_output:
            return out_loc, (f, h, w)
        out = sequence_shard.all_gather_tokens(out_loc, n)
    else:
        skip = tea_cache is not None and tea_cache.check(dit, x, TimeModulation(mod_rows_t, first_rows, n))
        out = yield from dit.forward_tokens_steps(x, ctx, mod_rows_t, t_rows, first_rows, (cos, sin), None, None, tea_cache, skip,
                                                  cfg_prefix if tea_cache is None else None, kv)
        if not gather_output:
            return out, (f, h, w)
    return dit.unpatchify(out, (f, h, w))

"""Model files -> modules: ``ModelConfig``, state-dict readers, key-hash identification, ``ModelPool``.

Mirror of the reference's load-time boundary for this path: ``diffsynth/core/loader/config.py:10-117``
(``ModelConfig`` fields and vram_config), ``core/loader/file.py:5-40,101-121`` (readers, ``hash_model_file``),
``core/loader/model.py:8-56`` (meta-init -> load_state_dict(assign=True) -> .to(dtype, device) -> eval) and
``models/model_loader.py:62-105`` (``ModelPool.auto_load_model`` / ``fetch_model``).  Only ``path=`` loading
is supported: the modelscope / huggingface download branch raises (no network by design).
"""
import hashlib
import importlib
import json
import os
from dataclasses import dataclass
from typing import Optional, Union

import torch
from safetensors import safe_open


@dataclass
class ModelConfig:
    path: Union[str, list] = None
    model_id: str = None
    origin_file_pattern: Union[str, list] = None
    download_source: str = None
    local_model_path: str = None
    skip_download: bool = None
    offload_device: Optional[Union[str, torch.device]] = None
    offload_dtype: Optional[torch.dtype] = None
    onload_device: Optional[Union[str, torch.device]] = None
    onload_dtype: Optional[torch.dtype] = None
    preparing_device: Optional[Union[str, torch.device]] = None
    preparing_dtype: Optional[torch.dtype] = None
    computation_device: Optional[Union[str, torch.device]] = None
    computation_dtype: Optional[torch.dtype] = None
    clear_parameters: bool = False

    def check_input(self):
        if self.path is None and self.model_id is None:
            raise ValueError('No valid model files. Please use `ModelConfig(path="xxx")` or '
                             '`ModelConfig(model_id="xxx/yyy", origin_file_pattern="zzz")`. '
                             "`skip_download=True` only supports the first one.")

    def download_if_necessary(self):
        self.check_input()
        if self.path is None:
            raise RuntimeError(f"offline: ModelConfig(model_id={self.model_id!r}, origin_file_pattern="
                               f"{self.origin_file_pattern!r}) would need a download; pass ModelConfig(path=...)")
        if isinstance(self.path, list) and len(self.path) == 1:
            self.path = self.path[0]

    def vram_config(self):
        return {k: getattr(self, k) for k in (
            "offload_device", "offload_dtype", "onload_device", "onload_dtype",
            "preparing_device", "preparing_dtype", "computation_device", "computation_dtype")}


# --------------------------------------------------------------------------------- state-dict files
def load_state_dict(file_path, torch_dtype=None, device="cpu"):
    if isinstance(file_path, list):
        sd = {}
        for one in file_path:
            sd.update(load_state_dict(one, torch_dtype, device))
        return sd
    if file_path.endswith(".safetensors"):
        sd = {}
        with safe_open(file_path, framework="pt", device=str(device)) as f:
            for k in f.keys():
                t = f.get_tensor(k)
                sd[k] = t.to(torch_dtype) if torch_dtype is not None else t
        return sd
    sd = torch.load(file_path, map_location=device, weights_only=True)
    if len(sd) == 1:
        for wrapper in ("state_dict", "module", "model_state"):
            if wrapper in sd:
                sd = sd[wrapper]
                break
    if torch_dtype is not None:
        sd = {k: (v.to(torch_dtype) if isinstance(v, torch.Tensor) else v) for k, v in sd.items()}
    return sd


def _keys_dict(file_path):
    """key -> shape list (nested dicts preserved), without materialising safetensors data."""
    if isinstance(file_path, list):
        out = {}
        for one in file_path:
            out.update(_keys_dict(one))
        return out
    if file_path.endswith(".safetensors"):
        with safe_open(file_path, framework="pt", device="cpu") as f:
            return {k: f.get_slice(k).get_shape() for k in f.keys()}

    def walk(d):
        return {k: (list(v.shape) if isinstance(v, torch.Tensor) else walk(v)) for k, v in d.items()}
    return walk(load_state_dict(file_path))


def _keys_string(keys_dict, with_shape=True):
    items = []
    for key, value in keys_dict.items():
        if not isinstance(key, str):
            continue
        if isinstance(value, dict):
            items.append(key + "|" + _keys_string(value, with_shape))
        else:
            if with_shape:
                items.append(key + ":" + "_".join(map(str, list(value))))
            items.append(key)
    items.sort()
    return ",".join(items)


def hash_keys_dict(keys_dict, with_shape=True):
    return hashlib.md5(_keys_string(keys_dict, with_shape).encode("UTF-8")).hexdigest()


def hash_model_file(path, with_shape=True):
    """md5 over the sorted 'key:shape' + 'key' strings (core/loader/file.py:101-121)."""
    return hash_keys_dict(_keys_dict(path), with_shape)


def hash_state_dict_keys(state_dict, with_shape=True):
    return hash_keys_dict({k: list(v.shape) for k, v in state_dict.items() if isinstance(v, torch.Tensor)}, with_shape)


# ------------------------------------------------------------------------------- model identification
def WanVideoVAEStateDictConverter(state_dict):
    """utils/state_dict_converters/wan_video_vae.py:1-7 — files store the inner VideoVAE38_ keys."""
    if "model_state" in state_dict:
        state_dict = state_dict["model_state"]
    return {"model." + k: v for k, v in state_dict.items()}


TI2V_5B_DIT_KWARGS = {
    "has_image_input": False, "patch_size": [1, 2, 2], "in_dim": 48, "dim": 3072, "ffn_dim": 14336, "freq_dim": 256,
    "text_dim": 4096, "out_dim": 48, "num_heads": 24, "num_layers": 30, "eps": 1e-06, "seperated_timestep": True,
    "require_clip_embedding": False, "require_vae_embedding": False, "fuse_vae_embedding_in_latents": True,
}

# The entries of diffsynth/configs/model_configs.py (:92-96, :289-302) that the TI2V-5B path needs.
MODEL_CONFIGS = [
    {
        "model_hash": "9c8818c2cbea55eca56c7b447df170da",       # models_t5_umt5-xxl-enc-bf16 (model_configs.py:92-96)
        "model_name": "wan_video_text_encoder",
        "model_class": "fairygen_amd.wan_video_text_encoder.WanTextEncoder",
    },
    {
        "model_hash": "1f5ab7703c6fc803fdded85ff040c316",
        "model_name": "wan_video_dit",
        "model_class": "fairygen_amd.wan_video_dit.WanModel",
        "extra_kwargs": TI2V_5B_DIT_KWARGS,
    },
    {
        "model_hash": "e1de6c02cdac79f8b739f4d3698cd216",
        "model_name": "wan_video_vae",
        "model_class": "fairygen_amd.wan_video_vae.WanVideoVAE38",
        "state_dict_converter": "fairygen_amd.loader.WanVideoVAEStateDictConverter",
    },
]


def register_model_config(entry):
    """Add a (hash -> class) entry, e.g. for reduced-size test checkpoints."""
    MODEL_CONFIGS[:] = [e for e in MODEL_CONFIGS if e["model_hash"] != entry["model_hash"]] + [entry]


def _import(qualified):
    mod, name = qualified.rsplit(".", 1)
    return getattr(importlib.import_module(mod), name)


def load_model(model_class, path, config=None, torch_dtype=torch.bfloat16, device="cpu", state_dict_converter=None):
    """core/loader/model.py:8-56 without the VRAM-management branch (offload is disabled on this path)."""
    with torch.device("meta"):
        model = model_class(**(config or {}))
    sd = load_state_dict(path, torch_dtype, device)
    sd = state_dict_converter(sd) if state_dict_converter is not None else dict(sd)
    model.load_state_dict(sd, assign=True)
    model = model.to(dtype=torch_dtype, device=device)
    return model.eval()


class ModelPool:
    def __init__(self):
        self.model, self.model_name, self.model_path = [], [], []

    def auto_load_model(self, path, vram_config=None, vram_limit=None, clear_parameters=False):
        print(f"Loading models from: {json.dumps(path, indent=4)}")
        vram_config = vram_config or {"computation_dtype": torch.bfloat16, "computation_device": "cpu"}
        if vram_config.get("offload_dtype") is not None and vram_config.get("offload_device") is not None:
            raise NotImplementedError("VRAM offload management is out of scope (288 GB HBM: models stay resident)")
        model_hash = hash_model_file(path)
        loaded = False
        for cfg in MODEL_CONFIGS:
            if cfg["model_hash"] != model_hash:
                continue
            converter = _import(cfg["state_dict_converter"]) if "state_dict_converter" in cfg else None
            # computation_dtype fp8 = the reference's fp8 Linear mode (AutoWrappedLinear.enable_fp8, core/vram/layers.py:312):
            # parameters stay bf16 in HBM, the blocks' GEMMs run through torch._scaled_mm
            comp = vram_config["computation_dtype"]
            fp8 = comp in (torch.float8_e4m3fn, torch.float8_e4m3fnuz)
            model = load_model(_import(cfg["model_class"]), path, cfg.get("extra_kwargs", {}),
                               torch.bfloat16 if fp8 else comp, vram_config["computation_device"], converter)
            if fp8:
                if not hasattr(model, "enable_fp8_linear"):
                    raise NotImplementedError(f"fp8 computation is only built for the DiT Linears, not for {cfg['model_name']}")
                model.enable_fp8_linear(comp)
            self.model.append(model)
            self.model_name.append(cfg["model_name"])
            self.model_path.append(path)
            info = {"model_name": cfg["model_name"], "model_class": cfg["model_class"], "extra_kwargs": cfg.get("extra_kwargs")}
            print(f"Loaded model: {json.dumps(info, indent=4)}")
            loaded = True
        if not loaded:
            raise ValueError(f"Cannot detect the model type. File: {path}. Model hash: {model_hash}")

    def fetch_model(self, model_name, index=None):
        found = [(m, p) for m, p, n in zip(self.model, self.model_path, self.model_name) if n == model_name]
        if not found:
            print(f"No {model_name} models available. This is not an error.")
            return None
        if len(found) == 1:
            print(f"Using {model_name} from {json.dumps(found[0][1], indent=4)}.")
            return found[0][0]
        models = [m for m, _ in found]
        if index is None:
            return models[0]
        return models[:index] if isinstance(index, int) else models

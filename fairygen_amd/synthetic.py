"""Deterministic random-init checkpoints with the reference's exact key/shape sets.

No real weights exist offline, so parity tests and bench.py use synthetic tensors:
keys are visited in sorted order with ONE generator; '*.weight' of Linear/Conv ~ N(0, 0.02^2), biases
~ N(0, 0.02^2), norm weights / gammas = 1 + N(0, 0.1^2), 'modulation' ~ N(0,1)/sqrt(dim)
(models/wan_video_dit.py:210,259), all rounded to bf16.  The key/shape md5 of the full-size dicts equals the
hashes diffsynth identifies the models by (configs/model_configs.py:289-302), so the saved files load
through ModelPool exactly like the real checkpoints.
"""
import torch
from safetensors.torch import save_file

from .loader import TI2V_5B_DIT_KWARGS
from .wan_video_dit import WanModel
from .wan_video_vae import WanVideoVAE38

TINY_DIT_KWARGS = dict(TI2V_5B_DIT_KWARGS, dim=256, ffn_dim=512, num_heads=2, num_layers=2, text_dim=128)


def _shapes(module):
    return {k: tuple(v.shape) for k, v in module.state_dict().items()}


def dit_shapes(kwargs=None):
    with torch.device("meta"):
        return _shapes(WanModel(**(kwargs or TI2V_5B_DIT_KWARGS)))


TINY_TEXT_KWARGS = dict(vocab=100, dim=128, dim_attn=128, dim_ffn=256, num_heads=2, num_layers=2, num_buckets=32)


def text_encoder_shapes(kwargs=None):
    from .wan_video_text_encoder import WanTextEncoder
    with torch.device("meta"):
        return _shapes(WanTextEncoder(**(kwargs or {})))


def vae_shapes(dec_dim=256, dim=160, with_prefix=True):
    with torch.device("meta"):
        shapes = _shapes(WanVideoVAE38(dim=dim, dec_dim=dec_dim))
    return shapes if with_prefix else {k[len("model."):]: v for k, v in shapes.items()}


def random_state_dict(shapes, seed=1234, dtype=torch.bfloat16, device="cpu", only_prefix=None):
    g = torch.Generator(device).manual_seed(seed)
    sd = {}
    for key in sorted(shapes):
        shape = shapes[key]
        x = torch.randn(shape, generator=g, device=device, dtype=torch.float32)
        if only_prefix is not None and not key.startswith(only_prefix):
            continue            # the generator still advanced: a filtered dict matches the full one
        leaf = key.rsplit(".", 1)[-1]
        if leaf == "modulation":
            x = x / shape[-1] ** 0.5
        elif leaf == "gamma" or (leaf == "weight" and len(shape) == 1):
            x = 1.0 + 0.1 * x
        else:
            x = 0.02 * x
        sd[key] = x.to(dtype)
    return sd


def random_lora(dit_shapes_, rank=32, seed=4321, dtype=torch.bfloat16, adapter=".default"):
    """Rank-r LoRA on q,k,v,o of both attentions and ffn.0/ffn.2 of every block (stage1_id.sh:15-16), in the
    key format the FairyGen trainer saves and merge_weights.py merges."""
    g = torch.Generator("cpu").manual_seed(seed)
    out = {}
    for key in sorted(dit_shapes_):
        parts = key.split(".")
        if parts[0] != "blocks" or parts[-1] != "weight" or len(dit_shapes_[key]) != 2:
            continue
        if not (parts[2] in ("self_attn", "cross_attn") and parts[3] in "qkvo" or parts[2] == "ffn"):
            continue
        base = key[: -len(".weight")]
        o, i = dit_shapes_[key]
        out[f"{base}.lora_A{adapter}.weight"] = (0.02 * torch.randn((rank, i), generator=g)).to(dtype)
        out[f"{base}.lora_B{adapter}.weight"] = (0.02 * torch.randn((o, rank), generator=g)).to(dtype)
    return out


def save_checkpoint(sd, path):
    save_file({k: v.contiguous() for k, v in sd.items()}, path)
    return path

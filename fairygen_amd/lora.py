"""LoRA state-dict normalisation and fuse-into-base (mirror of diffsynth/utils/lora/general.py:4-62).

Load-time only: ``W <- W + alpha * (B @ A)`` is one small GEMM per target Linear (``torch.mm``, hipBLASLt on
the device the pipeline lives on), exactly the reference's arithmetic; the hot loop then runs the same
kernels on different weights.  File format = merge_weights.py:19-45 ({A1, B1+B2}).
"""
import torch


class GeneralLoRALoader:
    def __init__(self, device="cpu", torch_dtype=torch.float32):
        self.device, self.torch_dtype = device, torch_dtype

    def get_name_dict(self, lora_state_dict):
        names = {}
        for key in lora_state_dict:
            down_tag, up_tag = ("lora_down", "lora_up") if ".lora_up." in key else ("lora_A", "lora_B")
            if up_tag not in key:
                continue
            parts = key.split(".")
            at = parts.index(up_tag)
            if len(parts) > at + 2:       # adapter-name segment such as ".default"
                del parts[at + 1]
            del parts[at]
            if parts[0] == "diffusion_model":
                del parts[0]
            del parts[-1]                 # trailing "weight"
            names[".".join(parts)] = (key, key.replace(up_tag, down_tag))
        return names

    def convert_state_dict(self, state_dict, suffix=".weight"):
        out = {}
        for name, (up_key, down_key) in self.get_name_dict(state_dict).items():
            out[name + f".lora_B{suffix}"] = state_dict[up_key]
            out[name + f".lora_A{suffix}"] = state_dict[down_key]
        return out

    def fuse_lora_to_base_model(self, model, state_dict, alpha=1.0):
        state_dict = self.convert_state_dict(state_dict)
        targets = {k[: -len(".lora_B.weight")] for k in state_dict if k.endswith(".lora_B.weight")}
        updated = 0
        for name, module in model.named_modules():
            if name not in targets:
                continue
            up = state_dict[name + ".lora_B.weight"].to(device=self.device, dtype=self.torch_dtype)
            down = state_dict[name + ".lora_A.weight"].to(device=self.device, dtype=self.torch_dtype)
            if up.dim() == 4:
                delta = alpha * torch.mm(up.squeeze(3).squeeze(2), down.squeeze(3).squeeze(2)).unsqueeze(2).unsqueeze(3)
            else:
                delta = alpha * torch.mm(up, down)
            base = module.state_dict()
            base["weight"] = base["weight"].to(device=self.device, dtype=self.torch_dtype) + delta
            module.load_state_dict(base)
            updated += 1
        if hasattr(model, "invalidate_fused"):
            model.invalidate_fused()
        print(f"{updated} tensors are fused by LoRA. Fused LoRA layers cannot be cleared by `pipe.clear_lora()`.")
        return updated


def merge_lora_weights(stage1, stage2):
    """merge_weights.py:19-45 on in-memory dicts: keep A1, B = B1 + B2 (missing B2 keeps B1)."""
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

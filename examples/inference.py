"""animation/inference.py of the reference, on fairygen_amd (same calls; local checkpoint paths instead of downloads).

    python examples/inference.py --weights /path/to/Wan2.2-TI2V-5B --tokenizer /path/to/google/umt5-xxl \\
        --lora ./lora-libraries/pig_walk/merged/200_400.safetensors --image ./data/pig_walk/shot/1.png --out ./outputs/1.mp4
"""
import argparse
import glob
import os
import sys
from pathlib import Path

import torch
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # the repo root: run from anywhere, no install
from fairygen_amd import ModelConfig, WanVideoPipeline, save_video

NEGATIVE = ("色调艳丽，过曝，静态，细节模糊不清，字幕，风格，作品，画作，画面，静止，整体发灰，最差质量，低质量，JPEG压缩残留，丑陋的，残缺的，多余的手指，"
            "画得不好的手部，画得不好的脸部，畸形的，毁容的，形态畸形的肢体，手指融合，静止不动的画面，杂乱的背景，三条腿，背景人很多，倒着走")


def build_pipeline(weights, tokenizer, lora=None, fp8=False):
    dit_cfg = dict(computation_dtype=torch.float8_e4m3fn) if fp8 else {}
    pipe = WanVideoPipeline.from_pretrained(
        torch_dtype=torch.bfloat16, device="cuda",
        model_configs=[
            ModelConfig(path=os.path.join(weights, "models_t5_umt5-xxl-enc-bf16.pth")),
            ModelConfig(path=sorted(glob.glob(os.path.join(weights, "diffusion_pytorch_model*.safetensors"))), **dit_cfg),
            ModelConfig(path=os.path.join(weights, "Wan2.2_VAE.pth")),
        ],
        tokenizer_config=ModelConfig(path=tokenizer))
    if lora:
        pipe.load_lora(pipe.dit, lora, alpha=1)
    return pipe


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--weights", required=True)
    ap.add_argument("--tokenizer", required=True)
    ap.add_argument("--lora")
    ap.add_argument("--image", required=True)
    ap.add_argument("--prompt", default="[p]_character_[w]_motion [p] walks towards the camera, filling the frame with a sense of movement.")
    ap.add_argument("--out", default="./outputs/1.mp4")
    ap.add_argument("--fp8", action="store_true", help="the reference's fp8 Linear mode for the DiT blocks")
    a = ap.parse_args()
    pipe = build_pipeline(a.weights, a.tokenizer, a.lora, a.fp8)
    image = Image.open(a.image).convert("RGB").resize((832, 480))
    Path(a.out).parent.mkdir(parents=True, exist_ok=True)
    video = pipe(prompt=a.prompt, negative_prompt=NEGATIVE, input_image=image, num_frames=81, seed=1, tiled=True)
    print("wrote", save_video(video, a.out, fps=15, quality=5))


if __name__ == "__main__":
    main()

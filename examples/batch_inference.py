"""animation/batch_inference.py of the reference, on fairygen_amd, spread over the GPUs of a node as replicas.

    torchrun --nproc-per-node 8 examples/batch_inference.py --weights ... --tokenizer ... --lora ... \\
        --input ./data/pig_walk/shot --output ./outputs/pig_walk --replica-size 2

Every `<name>.png` + `<name>.txt` pair of the input folder is one shot (the reference's convention); shot i is served by replica
i mod (world / replica_size); a 2-GPU replica runs one CFG branch per GPU.  Single process: plain sequential loop.
"""
import argparse
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))      # the repo root: run from anywhere, no install
from examples.inference import NEGATIVE, build_pipeline
from fairygen_amd.batch import ShotScheduler


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--weights", required=True)
    ap.add_argument("--tokenizer", required=True)
    ap.add_argument("--lora")
    ap.add_argument("--input", required=True)
    ap.add_argument("--output", required=True)
    ap.add_argument("--replica-size", type=int, default=2)
    ap.add_argument("--fp8", action="store_true")
    a = ap.parse_args()
    if "RANK" in os.environ:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl")
    sched = ShotScheduler(replica_size=a.replica_size)
    pipe = build_pipeline(a.weights, a.tokenizer, a.lora, a.fp8)
    done = sched.run_folder(pipe, a.input, a.output, negative_prompt=NEGATIVE, num_frames=81, seed=1, tiled=True)
    for name, path in done:
        if path:
            print(f"saved: {path}")
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

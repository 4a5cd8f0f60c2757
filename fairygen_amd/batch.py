"""Multi-shot serving: the loop of ``animation/batch_inference.py:27-56`` spread over the GPUs of a node.

The reference runs the shots of a folder one after the other on one GPU.  Shots are independent, so on N GPUs the
natural decomposition is REPLICAS: the ranks are cut into disjoint sets of ``replica_size``
GPUs, every set runs whole clips with its own ``ParallelLayout`` (2 GPUs per clip = one CFG branch each, no exchange
inside the forward), and shot ``i`` goes to replica ``i % n_replicas``.  No collective crosses replicas.

One process per GPU (``torchrun``); every process calls ``ShotScheduler(...)`` with the same arguments, because creating
the replicas' process groups is collective.
"""
import os

import torch.distributed as dist

from .sequence_parallel import ParallelLayout, replica_ranks


def list_shots(input_folder):
    """batch_inference.py:24-39: every ``<name>.png`` with a ``<name>.txt`` prompt next to it, sorted by file name;
    images without a prompt are skipped (reported), as in the reference."""
    shots = []
    for img_name in sorted(f for f in os.listdir(input_folder) if f.endswith(".png")):
        base = os.path.splitext(img_name)[0]
        txt_path = os.path.join(input_folder, base + ".txt")
        if not os.path.exists(txt_path):
            print(f"Skip: can not find corresponding txt file: {base}.txt")
            continue
        shots.append((base, os.path.join(input_folder, img_name), txt_path))
    return shots


class ShotScheduler:
    def __init__(self, replica_size=2, cfg_parallel=None, attn_mode="ulysses"):
        """replica_size GPUs per clip; cfg_parallel defaults to 2 when the replica size is even (a 2-GPU replica then
        has no exchange inside the forward), else 1."""
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        rank = dist.get_rank() if world > 1 else 0
        replica_size = min(replica_size, world)
        if cfg_parallel is None:
            cfg_parallel = 2 if replica_size % 2 == 0 else 1
        self.replicas = replica_ranks(world, replica_size)
        layouts = [ParallelLayout(cfg_parallel, attn_mode, ranks=r) if world > 1 else None for r in self.replicas]
        self.replica_id = next(i for i, r in enumerate(self.replicas) if rank in r)
        self.layout = layouts[self.replica_id]
        self.is_writer = rank == self.replicas[self.replica_id][0]      # one rank per replica writes the outputs

    @property
    def n_replicas(self):
        return len(self.replicas)

    def attach(self, pipe):
        """Make `pipe` (this process's pipeline) work inside its replica."""
        if self.layout is not None and len(self.layout.ranks) > 1:
            pipe.enable_sequence_parallel(layout=self.layout)
        return pipe

    def my_shots(self, shots):
        """The shots this process's replica serves: round-robin in the reference's (sorted) order."""
        return [s for i, s in enumerate(shots) if i % self.n_replicas == self.replica_id]

    def run_folder(self, pipe, input_folder, output_folder, negative_prompt="", size=(832, 480), fps=15, quality=5,
                   **call_kwargs):
        """batch_inference.py end to end for this process's share: open image + prompt, pipe(...), save_video.
        Returns [(shot name, output path)] (paths only on the replica's writer rank)."""
        from PIL import Image
        from .data import save_video
        self.attach(pipe)
        os.makedirs(output_folder, exist_ok=True)
        done = []
        for base, img_path, txt_path in self.my_shots(list_shots(input_folder)):
            image = Image.open(img_path).convert("RGB").resize(size)
            with open(txt_path, "r", encoding="utf-8") as f:
                prompt = f.read().strip()
            video = pipe(prompt=prompt, negative_prompt=negative_prompt, input_image=image, **call_kwargs)
            out = save_video(video, os.path.join(output_folder, f"{base}.mp4"), fps=fps, quality=quality) if self.is_writer else None
            done.append((base, out))
        return done

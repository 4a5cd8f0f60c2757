"""bench.py — FairyGen animation hot path on MI355X: decoded frames/sec and sec/clip.

    python bench.py [--gpus N] [--steps K] [--warmup W]            (N>1: launched by torch.distributed.run)

Workload (BASELINE.json metric / configs[2]): Wan2.2-TI2V-5B, 704x1280x121, TI2V (first latent frame pinned),
CFG 5.0, sigma shift 5.0, merged rank-32 motion LoRA fused at load, random-init bf16 weights with the reference's
key/shape set, synthetic context tensors.  A "step" is one denoise step of the clip =
forward(+), forward(-), CFG combine + Euler update, first-frame re-pin.  The timed region is the WHOLE clip for
num_inference_steps = K: K steps followed by the VAE decode to (1,3,121,704,1280) (tiled exactly like
inference.py's tiled=True unless --untiled), inputs resident in HBM; value = frames / t_clip.  Default K = 50
is the headline configuration (BASELINE.json: "@ 50 steps"); ms_per_step = t_clip / K (decode amortised).  The `metric`
string names the K that was actually run; whatever K the caller passes, config.frames_per_s_at_50_steps carries the
BASELINE metric: measured directly when K = 50, otherwise derived as 121 / (50 x measured denoise step + measured decode)
and labelled so.  `roofline` describes the dominant kernel (self-attention) and lists, under "kernels", the other MFMA
consumers of the clip measured with HIP events on the launch stream: the VAE conv kernels (MFMA fraction and algorithmic
HBM GB/s of the decode) and the hipBLASLt GEMMs of the DiT blocks.
N > 1 ("strong" scaling: one clip, total work fixed): world = cfg_parallel x sp (fairygen_amd/sequence_parallel.py):
tokens sharded by latent-temporal ranges inside a sequence-parallel group with either an RCCL K/V all-gather or the
Ulysses all-to-all pair around every self-attention, optionally one CFG branch per half of the ranks; VAE tiles
are dealt over all ranks.  --layout auto (default) times one denoise step of every candidate layout during the
untimed warm-up and keeps the fastest (env FAIRYGEN_PARALLEL=cfg2-ulysses etc. forces one).
Launched by torch.distributed.run with ONE rank (RANK / WORLD_SIZE in the env) the process group is still an RCCL
communicator: barrier, MAX all-reduce and — with FAIRYGEN_FORCE_COLLECTIVES=1 and a forced --layout — every exchange of the
sharded path execute on it (single-GPU rehearsal of the N > 1 code path; the line says so in config.parallelism).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA, MI355X (/opt/skills/guides/MI355X_MICROARCH.md)
PEAK_FP8_TFLOPS = 5000.0       # dense fp8 MFMA (block-scaled f8f6f4 form), same guide
# HBM bytes per self-attention launch from rocprofv3 PMC passes of the SAME kernel and shape (profiles/r02_attn_w4_pmc.json:
# separate --pmc FETCH_SIZE / WRITE_SIZE passes, KB units, gfx950 x2 correction on FETCH_SIZE), keyed by (Nq, Nkv, H).
# PMC collection cannot run inside the timed bench; other shapes report null.
PMC_HBM_BYTES_PER_LAUNCH = {(27280, 27280, 24): 2.110e9}


def seeded(shape, seed, dtype=torch.bfloat16):
    g = torch.Generator("cpu").manual_seed(seed)
    return torch.randn(shape, generator=g, dtype=torch.float32).to(dtype)


def build_pipeline(args, device):
    from fairygen_amd import synthetic
    from fairygen_amd.loader import TI2V_5B_DIT_KWARGS
    from fairygen_amd.wan_video import WanVideoPipeline
    from fairygen_amd.wan_video_dit import WanModel
    from fairygen_amd.wan_video_vae import WanVideoVAE38

    cfg = dict(TI2V_5B_DIT_KWARGS)
    if args.layers:
        cfg["num_layers"] = args.layers           # debugging only; the JSON line then says so
    pipe = WanVideoPipeline(device=device, torch_dtype=torch.bfloat16)
    shapes = synthetic.dit_shapes(cfg)
    with torch.device("meta"):
        dit = WanModel(**cfg)
    dit.load_state_dict(synthetic.random_state_dict(shapes, seed=1234, device=device), assign=True)
    pipe.dit = dit.to(device=device, dtype=torch.bfloat16).eval()
    if not args.no_lora:
        lora = synthetic.random_lora(shapes, rank=32, seed=4321)
        pipe.load_lora(pipe.dit, state_dict=lora, alpha=1)
    with torch.device("meta"):
        vae = WanVideoVAE38()
    vae.load_state_dict(synthetic.random_state_dict(synthetic.vae_shapes(), seed=1234, device=device, only_prefix="model.dec")
                        | synthetic.random_state_dict({k: v for k, v in synthetic.vae_shapes().items() if not k.startswith("model.dec")},
                                                      seed=1, device=device), assign=True)
    pipe.vae = vae.to(device=device, dtype=torch.bfloat16).eval()
    pipe.height_division_factor = pipe.width_division_factor = 32
    return pipe, cfg


class KernelTimer:
    """HIP events (on torch's current stream = the stream the kernels are launched on) around every launch of the timed
    region of: fg_attn_fwd_bf16 (hip.attention), fg_conv3d_cl_bf16 (hip.conv3d_cl) and the hipBLASLt bias-GEMMs of the DiT
    blocks (wan_video_dit.gemm_bias*)."""

    def __init__(self):
        self.attn, self.conv, self.gemm, self.gemm_own, self.gemm_fp8 = [], [], [], [], []

    @staticmethod
    def _events():
        return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def install(self):
        from fairygen_amd import hip, wan_video_dit
        self._orig = {"attention": hip.attention, "conv3d_cl": hip.conv3d_cl}
        self._orig_gemm = {n: getattr(wan_video_dit, n) for n in ("gemm_bias", "gemm_bias_gelu", "gemm_bias_tuned", "gemm_bias_own",
                                                                   "gemm_bias_gelu_own", "gemm_residual", "gemm_fp8_own")}
        timer, lib = self, hip.load()

        def timed_attention(q, k, v, num_heads, out=None, scale=None):
            s, e = timer._events()
            s.record()
            r = timer._orig["attention"](q, k, v, num_heads, out, scale)
            e.record()
            timer.attn.append((q.shape[1], k.shape[1], num_heads, s, e))
            return r

        def timed_conv(x, w_packed, bias, cout, kt, ks, residual=None, **kw):
            s, e = timer._events()
            s.record()
            r = timer._orig["conv3d_cl"](x, w_packed, bias, cout, kt, ks, residual=residual, **kw)
            e.record()
            cin, pixels = x.shape[3], r.numel() // cout
            t, h, w = (r.shape[0] // 2, r.shape[1], r.shape[2]) if kw.get("time_interleave") else r.shape[:3]
            nbytes = 2 * (x.numel() + r.numel() * (2 if residual is not None else 1) + cout * cin * kt * ks * ks)
            timer.conv.append((lib.fg_conv_tile_choice(t, h, w, cout), 2.0 * pixels * cout * cin * kt * ks * ks, nbytes, s, e))
            return r

        def timed_gemm(name):
            def fn(x, weight, bias):
                s, e = timer._events()
                s.record()
                r = timer._orig_gemm[name](x, weight, bias)
                e.record()
                (timer.gemm_own if name.endswith("_own") else timer.gemm).append((x.numel() // x.shape[-1], weight.shape[1], weight.shape[0], s, e))
                return r

            def fn_fp8(xq, scale_a, w8, bias):
                s, e = timer._events()
                s.record()
                r = timer._orig_gemm[name](xq, scale_a, w8, bias)
                e.record()
                timer.gemm_fp8.append((xq.shape[0], w8.shape[1], w8.shape[0], s, e))
                return r

            def fn_residual(x, a, weight, bias, mod=None, gate_idx=None):
                s, e = timer._events()
                s.record()
                r = timer._orig_gemm[name](x, a, weight, bias, mod, gate_idx)
                e.record()
                timer.gemm_own.append((a.numel() // a.shape[-1], weight.shape[1], weight.shape[0], s, e))
                return r
            return fn_residual if name == "gemm_residual" else fn_fp8 if name == "gemm_fp8_own" else fn
        hip.attention, hip.conv3d_cl = timed_attention, timed_conv
        for n in self._orig_gemm:
            setattr(wan_video_dit, n, timed_gemm(n))

    def uninstall(self):
        from fairygen_amd import hip, wan_video_dit
        hip.attention, hip.conv3d_cl = self._orig["attention"], self._orig["conv3d_cl"]
        for n, f in self._orig_gemm.items():
            setattr(wan_video_dit, n, f)

    def self_attention_stats(self):
        durs, flops, shape = [], 0.0, None
        for nq, nkv, h, s, e in self.attn:
            if nkv > 1024:         # self-attention launches (cross-attention has Nkv = 512)
                durs.append(s.elapsed_time(e) * 1e-3)
                flops, shape = 4.0 * nq * nkv * h * 128, (nq, nkv, h)
        if not durs:
            return None
        avg = sum(durs) / len(durs)
        return {"launches": len(durs), "avg_s": avg, "flops_per_launch": flops, "tflops": flops / avg / 1e12,
                "traffic": PMC_HBM_BYTES_PER_LAUNCH.get(shape)}

    def other_kernels(self):
        """Roofline entries of the other MFMA consumers: aggregate FLOPs / aggregate event time over the timed region."""
        out = []
        x_t = sum(s.elapsed_time(e) for nq, nkv, h, s, e in self.attn if nkv <= 1024) * 1e-3
        x_f = sum(4.0 * nq * nkv * h * 128 for nq, nkv, h, s, e in self.attn if nkv <= 1024)
        if x_t > 0:
            out.append({"kernel": "attn_fwd_kernel<8,1,short-KV> (cross-attention, 512 keys)", "bound": "mfma",
                        "achieved": round(x_f / x_t / 1e12, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(x_f / x_t / 1e12 / PEAK_BF16_TFLOPS, 4), "total_s": round(x_t, 3)})
        for variant, name in ((256, "conv3d_cl_w4_kernel / conv3d_cl_256p_kernel (VAE38 decode, 256x256x64 LDS-DMA tile; hand-scheduled 4-wave form where Cin % 64 == 0)"),
                              (128, "conv3d_cl_kernel (VAE38 decode, 128x128x64 tile: low-resolution / odd-channel layers)")):
            rec = [(f, b, s.elapsed_time(e) * 1e-3) for v, f, b, s, e in self.conv if v == variant]
            if rec:
                t = sum(r[2] for r in rec)
                fl, by = sum(r[0] for r in rec), sum(r[1] for r in rec)
                out.append({"kernel": name, "bound": "mfma", "achieved": round(fl / t / 1e12, 1), "peak": PEAK_BF16_TFLOPS,
                            "unit": "TFLOP/s", "frac": round(fl / t / 1e12 / PEAK_BF16_TFLOPS, 4), "launches": len(rec),
                            "total_s": round(t, 3), "algorithmic_hbm_GBps": round(by / t / 1e9, 1),
                            "hbm_frac_of_8TBps": round(by / t / 8e12, 4)})
        rec = [(2.0 * m * k * n, s.elapsed_time(e) * 1e-3) for m, k, n, s, e in self.gemm if m >= 1024]
        if rec:
            t, fl = sum(r[1] for r in rec), sum(r[0] for r in rec)
            out.append({"kernel": "hipBLASLt bias-GEMMs of the DiT blocks (library: those not taken by gemm_p_kernel)",
                        "bound": "mfma", "achieved": round(fl / t / 1e12, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(fl / t / 1e12 / PEAK_BF16_TFLOPS, 4), "launches": len(rec), "total_s": round(t, 3)})
        rec = [(2.0 * m * k * n, s.elapsed_time(e) * 1e-3) for m, k, n, s, e in self.gemm_own]
        if rec:
            t, fl = sum(r[1] for r in rec), sum(r[0] for r in rec)
            out.append({"kernel": "gemm_p_kernel<bf16> (fg_gemm_epilogue_bf16: persistent hand-scheduled DiT GEMM; qkv, cross q, ffn.0 with GELU in "
                                  "the store, o / cross-o / ffn.2 with the gate*y + x residual in the store)",
                        "bound": "mfma", "achieved": round(fl / t / 1e12, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(fl / t / 1e12 / PEAK_BF16_TFLOPS, 4), "launches": len(rec), "total_s": round(t, 3)})
        rec = [(2.0 * m * k * n, s.elapsed_time(e) * 1e-3) for m, k, n, s, e in self.gemm_fp8 if m >= 1024]
        if rec:
            t, fl = sum(r[1] for r in rec), sum(r[0] for r in rec)
            out.append({"kernel": "gemm_p_kernel<e4m3> (fg_gemm_fp8_bf16: the fp8 Linear mode's row-scaled matmul, v_mfma_f32_32x32x64_f8f6f4)",
                        "bound": "mfma", "achieved": round(fl / t / 1e12, 1), "peak": PEAK_FP8_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(fl / t / 1e12 / PEAK_FP8_TFLOPS, 4), "launches": len(rec), "total_s": round(t, 3)})
        return out


LAYOUTS = ("cfg1-allgather", "cfg1-ulysses", "cfg2-allgather", "cfg2-ulysses", "cfg1-windows")


def candidate_layouts(world, num_heads):
    """(cfg_parallel, attn_mode) candidates for a world size; the attention exchange is moot when sp == 1."""
    out = []
    for cfgp in (2, 1):
        if world % cfgp:
            continue
        sp = world // cfgp
        if sp == 1:
            out.append((cfgp, "allgather"))
            continue
        if num_heads % sp == 0:
            out.append((cfgp, "ulysses"))
        out.append((cfgp, "allgather"))
    return out


def cpu_baseline(args, n_tokens, frames, steps, grid):
    """The CPU oracle ("port" of the reference's PyTorch path, pinned bit-exact to it by tests/golden — at full model width by
    oracle/gen_config1.py, where the reference itself took 25.0 s/clip and this port 24.1 s on config 1, and at the headline
    N = 27 280 by oracle/gen_config3_forward.py) timed on the host cores on a bounded sample of the SAME workload: ONE of the
    30 full-width DiT blocks at the REAL token count (no N^2 extrapolation: x30 blocks x2 CFG branches x steps is a plain
    repeat count) + a full-width VAE38 decode of a (1,48,2,12,12) latent scaled by pixel-frames."""
    from fairygen_amd import synthetic
    from fairygen_amd.loader import TI2V_5B_DIT_KWARGS
    from oracle import wan_dit, wan_vae
    cfg = dict(TI2V_5B_DIT_KWARGS, num_layers=1)
    shapes = {k: v for k, v in synthetic.dit_shapes(cfg).items() if k.startswith("blocks.0.")}
    sd = synthetic.random_state_dict(shapes, seed=5)
    ns = args.cpu_tokens or n_tokens
    f, h, w = grid if ns == n_tokens else (1, 64, ns // 64)
    x, ctx = seeded((1, ns, 3072), 6), seeded((1, 512, 3072), 7)
    t_mod = seeded((1, 1, 6, 3072), 8).expand(1, ns, 6, 3072).contiguous()      # per-token, as the reference materialises it in TI2V mode
    table = wan_dit.rope_table_3d(128, f, h, w)
    with torch.no_grad():
        t0 = time.perf_counter()
        wan_dit.dit_block(sd, "blocks.0", x, ctx, t_mod, table, 24, 1e-6)
        t_block = time.perf_counter() - t0
        if ns == n_tokens:
            t_forward, how = 30 * t_block, f"1 of 30 full-width DiT blocks at the real N={ns} tokens ({t_block:.1f}s)"
        else:           # --cpu-tokens: a shorter sample, scaled (GEMM ~ N, SDPA ~ N^2) — not the default
            q = seeded((1, ns, 3072), 9)
            t0 = time.perf_counter()
            wan_dit.attention(q, q, q, 24)
            t_attn = time.perf_counter() - t0
            r = n_tokens / ns
            t_forward = 30 * ((t_block - t_attn) * r + t_attn * r * r)
            how = f"1 of 30 full-width DiT blocks at {ns} tokens ({t_block:.1f}s, SDPA {t_attn:.1f}s) scaled to N={n_tokens} (GEMM~N, SDPA~N^2)"
        vsd = synthetic.random_state_dict(synthetic.vae_shapes(), seed=1234, only_prefix="model.dec") \
            | synthetic.random_state_dict({"model.conv2.weight": (48, 48, 1, 1, 1), "model.conv2.bias": (48,)}, seed=2)
        # a (2, 12, 12) latent = 5 frames of 192 x 192: big enough that the per-op overhead of a tiny tensor on 128 threads no longer
        # dominates (a (2, 4, 4) sample cost 4x more per pixel-frame and swung between 3 and 7 s from run to run); timed after a tiny
        # warm-up call (thread pools, lazy initialisation)
        wan_vae.decode(vsd, seeded((1, 48, 1, 2, 2), 11))
        z = seeded((1, 48, 2, 12, 12), 10)
        t0 = time.perf_counter()
        wan_vae.decode(vsd, z)
        t_vae_s = time.perf_counter() - t0
    lat_t = (frames - 1) // 4 + 1
    vae_scale = (args.height // 16) * (args.width // 16) * (4 * lat_t - 3) / (12 * 12 * 5)
    t_clip = t_forward * 2 * steps + t_vae_s * vae_scale * (2.21 if not args.untiled and args.height > 480 else 1.0)
    return {"value": frames / t_clip, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{how} x30 x2 x{steps} steps + full-width VAE38 decode of a (1,48,2,12,12) latent ({t_vae_s:.1f}s) "
                      f"scaled by pixel-frames; sec/clip = {t_clip:.0f}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--height", type=int, default=704)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--frames", type=int, default=121)
    ap.add_argument("--untiled", action="store_true", help="single_decode instead of the reference's tiled=True decode")
    ap.add_argument("--no-lora", action="store_true")
    ap.add_argument("--layers", type=int, default=0, help="debug: fewer DiT layers (reported in config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-tokens", type=int, default=0, help="cpu_baseline sample: DiT block token count (0 = the real N, no scaling)")
    ap.add_argument("--skip-vae", action="store_true", help="debug: denoise loop only (reported in config)")
    ap.add_argument("--gelu-epilogue", type=int, default=1, help="0: separate GELU kernel after ffn.0 instead of the GEMM epilogue")
    ap.add_argument("--linear-dtype", default="bf16", choices=("bf16", "fp8"),
                    help="fp8: the reference's fp8 Linear mode for the DiT blocks (config 5's weight path); NOT the headline config")
    ap.add_argument("--sliding-window", default="", metavar="SIZE,STRIDE",
                    help="the reference's APPROXIMATE sliding-window mode (latent frames per window, stride); with --layout "
                         "cfg1-windows the windows are dealt to the ranks.  Never the headline configuration")
    ap.add_argument("--gemm-grid", type=int, default=0, help="experiment: launch the persistent GEMM with this many workgroups (fg_gemm_debug_grid; reported in config)")
    ap.add_argument("--layout", default=os.environ.get("FAIRYGEN_PARALLEL", "auto"), choices=("auto",) + LAYOUTS,
                    help="N>1: how the ranks are used (auto: time every candidate for one step in the warm-up, keep the fastest)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    if args.gemm_grid:
        from fairygen_amd import hip as _hip
        _hip.load().fg_gemm_debug_grid(args.gemm_grid)
    distributed = world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ)      # launched by torch.distributed.run
    if distributed:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("FAIRYGEN_BENCH_BACKEND", "nccl")      # "gloo": rehearsal with several ranks on one GPU
        if backend == "nccl":
            dist.init_process_group(backend=backend, device_id=torch.device(device))
        else:
            dist.init_process_group(backend=backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    from fairygen_amd.sequence_parallel import force_collectives
    rehearsal = distributed and world == 1 and force_collectives()      # 1-rank RCCL communicator running every exchange

    from fairygen_amd import hip
    hip.load()
    pipe, cfg = build_pipeline(args, device)
    pipe.dit.gelu_epilogue = bool(args.gelu_epilogue)
    if args.linear_dtype == "fp8":
        pipe.dit.enable_fp8_linear(torch.float8_e4m3fn)
    if args.layout == "cfg1-windows" and not args.sliding_window:
        ap.error("--layout cfg1-windows needs --sliding-window SIZE,STRIDE")
    if (world > 1 or rehearsal) and args.layout != "auto":
        cfgp, mode = args.layout.split("-")
        pipe.enable_sequence_parallel(cfg_parallel=int(cfgp[3:]), attn_mode=mode)
    H, W, F_ = args.height, args.width, args.frames
    lat_shape = (1, 48, (F_ - 1) // 4 + 1, H // 16, W // 16)
    n_tokens = lat_shape[2] * (lat_shape[3] // 2) * (lat_shape[4] // 2)
    noise = seeded(lat_shape, 1).to(device)
    ctx_p = seeded((1, 512, 4096), 2); ctx_p[:, 64:] = 0
    ctx_n = seeded((1, 512, 4096), 3); ctx_n[:, 128:] = 0
    ctx_p, ctx_n = ctx_p.to(device), ctx_n.to(device)
    z0 = seeded((1, 48, 1, lat_shape[3], lat_shape[4]), 4).to(device)

    def run_clip(steps, decode):
        pipe.scheduler.set_timesteps(steps, denoising_strength=1.0, shift=5.0)
        latents = noise.clone()
        latents[:, :, 0:1] = z0
        shared = {"latents": latents, "fuse_vae_embedding_in_latents": True, "first_frame_latents": z0}
        if args.sliding_window:
            size, stride = (int(v) for v in args.sliding_window.split(","))
            shared.update(sliding_window_size=size, sliding_window_stride=stride)
        latents = pipe.denoise(shared, {"context": ctx_p}, {"context": ctx_n}, 5.0, progress_bar_cmd=lambda x: x)
        if not decode:
            return latents, None
        torch.cuda.synchronize()
        phase["denoise_s"] = time.perf_counter() - phase["t0"]
        video = pipe.decode_latents(latents, tiled=not args.untiled)
        return latents, video

    def barrier():
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[local_rank])
        else:
            dist.barrier()

    def sync_max(t):
        if not distributed:
            return t
        tt = torch.tensor([t], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return tt.item()

    phase = {}
    autotune = None
    with torch.no_grad():
        if world > 1 and args.layout == "auto":
            # untimed: one warm step + two timed steps of every candidate layout at full shape; all ranks agree on the
            # fastest through a MAX all-reduce of their times
            autotune = {}
            for cfgp, mode in candidate_layouts(world, cfg["num_heads"]):
                pipe.enable_sequence_parallel(cfg_parallel=cfgp, attn_mode=mode)
                run_clip(1, decode=False)
                torch.cuda.synchronize()
                barrier()
                t0 = time.perf_counter()
                run_clip(2, decode=False)
                torch.cuda.synchronize()
                autotune[f"cfg{cfgp}-{mode}"] = round(sync_max(time.perf_counter() - t0) / 2 * 1e3, 2)
            best = min(autotune, key=autotune.get)
            cfgp, mode = best.split("-")
            pipe.enable_sequence_parallel(cfg_parallel=int(cfgp[3:]), attn_mode=mode)
        # warmup: W denoise steps at full shape + a small decode (packs the VAE weights, warms the allocator)
        if args.warmup > 0:
            run_clip(args.warmup, decode=False)
        if not args.skip_vae:
            pipe.vae.decode(noise[:, :, :2, :8, :8].contiguous(), device=device, tiled=False)
            if pipe.parallel is not None and pipe.parallel.world.active:      # first-use set-up of the tile broadcast stays untimed
                pipe.parallel.world.broadcast(torch.zeros(1024, dtype=torch.bfloat16, device=device), src=0)
        torch.cuda.synchronize()
        timer = KernelTimer()
        timer.install()
        if distributed:
            barrier()
        torch.cuda.synchronize()
        t0 = phase["t0"] = time.perf_counter()
        latents, video = run_clip(args.steps, decode=not args.skip_vae)
        torch.cuda.synchronize()
        if distributed:
            barrier()
        torch.cuda.synchronize()
        t_clip = time.perf_counter() - t0
        timer.uninstall()
    t_clip = sync_max(t_clip)

    if rank == 0:
        st = timer.self_attention_stats()
        roofline = None
        if st:
            roofline = {"bound": "mfma", "kernel": "attn_fwd_w4_kernel (fg_attn_fwd_bf16, self-attention launches: main kernel + split-KV merge)",
                        "achieved": round(st["tflops"], 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(st["tflops"] / PEAK_BF16_TFLOPS, 4), "traffic": st["traffic"],
                        "launches": st["launches"], "avg_launch_ms": round(st["avg_s"] * 1e3, 3),
                        "flops_per_launch": st["flops_per_launch"], "kernels": timer.other_kernels()}
        denoise_s = phase.get("denoise_s", t_clip)
        decode_s = t_clip - denoise_s
        headline = (H, W, F_) == (704, 1280, 121) and not args.untiled and not args.skip_vae and not args.layers \
            and not args.sliding_window and args.linear_dtype == "bf16"
        t50 = t_clip if args.steps == 50 else 50 * denoise_s / args.steps + decode_s
        line = {
            "metric": f"decoded frames/sec (sec/clip in config), Wan2.2-TI2V-5B {H}x{W}x{F_} @ {args.steps} steps",
            "value": round(F_ / t_clip, 4), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(t_clip / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "bf16" if args.linear_dtype == "bf16" else "fp8_e4m3fn block linears (reference fp8_linear), bf16 elsewhere",
            "data": "synthetic",
            "config": {"workload": f"Wan2.2-TI2V-5B {H}x{W}x{F_} TI2V clip: {args.steps} denoise steps (CFG 5.0, 2 forwards/step) + "
                                   f"{'untiled' if args.untiled else 'tiled (30,52)/(15,26)'} VAE38 decode",
                       "headline_config": bool(headline and args.steps == 50),
                       "frames_per_s_at_50_steps": round(F_ / t50, 4), "sec_per_clip_at_50_steps": round(t50, 2),
                       "at_50_steps_is": "measured" if args.steps == 50 else
                                         f"derived: 50 x measured denoise step ({denoise_s / args.steps * 1e3:.1f} ms) + measured decode ({decode_s:.2f} s)",
                       "sec_per_clip": round(t_clip, 2), "denoise_s": round(denoise_s, 2),
                       "denoise_ms_per_step": round(denoise_s / args.steps * 1e3, 2),
                       "vae_decode_s": round(decode_s, 2), "tokens": n_tokens, "num_inference_steps": args.steps,
                       "lora": "rank-32 merged, fused at load" if not args.no_lora else "none",
                       "parallelism": (pipe.parallel.describe() + (" REHEARSAL on a 1-rank RCCL communicator (FAIRYGEN_FORCE_COLLECTIVES=1)"
                                                                    if rehearsal else "")) if pipe.parallel is not None else
                                      ("single" + (" (1-rank RCCL process group: barrier + MAX all-reduce)" if distributed else "")),
                       "weights": "random-init bf16, reference key/shape set"},
            "roofline": roofline,
        }
        from fairygen_amd import wan_video as _wv, wan_video_dit as _wd
        line["config"]["gelu"] = ("gemm_p_kernel epilogue (mode 4)" if _wd.GEMM_BACKEND == "all" and args.linear_dtype == "bf16" else
                                  "fused into ffn.2's row quantisation" if args.linear_dtype == "fp8" else
                                  "hipBLASLt epilogue" if args.gelu_epilogue else "fg_act_bf16 kernel")
        line["config"]["dit_gemm_backend"] = _wd.GEMM_BACKEND + " (FAIRYGEN_GEMM: which Linears run on gemm_p_kernel instead of hipBLASLt; 'all' = every Linear of the blocks)"
        if args.gemm_grid:
            line["config"]["EXPERIMENT_gemm_workgroups"] = args.gemm_grid
        if args.linear_dtype == "fp8":
            line["config"]["fp8_gemm"] = _wd.FP8_GEMM + " (FAIRYGEN_FP8_GEMM: own = fg_gemm_fp8_bf16, lib = torch._scaled_mm)"
        line["config"]["attention_scale"] = ("2^-3 / log2(e) with 1.0201 folded into q's RoPE table: exact pre-multiplied form of attn_fwd_w4_kernel"
                                             if pipe.dit.attn_scale()[0] is not None else "1/sqrt(d): plain form of attn_fwd_w4_kernel")
        line["config"]["cfg_shared_prefix"] = ("block 0's self-attention computed once per step for both CFG forwards (identical inputs, bit-identical "
                                               "result; FAIRYGEN_CFG_SHARE=0 computes it twice)") if _wv.CFG_SHARE_PREFIX else "off"
        line["config"]["cross_attention_kv"] = ("computed at the first step, kept for the loop (prompt and weights are constant; bit-identical; "
                                                "FAIRYGEN_CROSS_KV_CACHE=0 recomputes per step)") if _wv.CROSS_KV_CACHE else "recomputed per step"
        if args.sliding_window:
            line["config"]["APPROXIMATE_sliding_window"] = args.sliding_window + " (reference TemporalTiler mode, not the default path)"
        if autotune is not None:
            line["config"]["layout_autotune_ms_per_step"] = autotune
        if args.layers:
            line["config"]["DEBUG_num_layers"] = args.layers
        if args.skip_vae:
            line["config"]["DEBUG_skip_vae"] = True
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args, n_tokens, F_, args.steps, (lat_shape[2], lat_shape[3] // 2, lat_shape[4] // 2))
        print(json.dumps(line))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

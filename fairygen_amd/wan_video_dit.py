"""Wan2.2-TI2V-5B DiT on MI355X: parameter layout of the reference, compute on the HIP kernels.

Mirror of ``diffsynth/models/wan_video_dit.py`` (``WanModel`` :271-336, ``DiTBlock`` :195-229,
``SelfAttention`` :123-146, ``CrossAttention`` :149-185, ``Head`` :252-268): same constructor kwargs,
same parameter names and shapes, so reference checkpoints, LoRA files and the key-hash model
identification (``core/loader/file.py:117-121``) keep working.  The modules hold parameters only; the
arithmetic is in ``forward_tokens`` below, built from ``fairygen_amd.hip`` kernels plus hipBLASLt GEMMs
(``F.linear``).  Nothing here runs on CPU tensors — the HIP library raises.
"""
import math

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import hip, tuning


def sinusoidal_embedding_1d(dim, position):
    """fp64 sinusoid [cos | sin] (reference :67-71); position is a small CPU tensor here."""
    sinusoid = torch.outer(position.to(torch.float64),
                           torch.pow(10000.0, -torch.arange(dim // 2, dtype=torch.float64).div(dim // 2)))
    return torch.cat([sinusoid.cos(), sinusoid.sin()], dim=1).to(position.dtype)


def precompute_freqs_cis(dim, end=1024, theta=10000.0):
    """complex128 rotation table of one axis (reference :82-88)."""
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 2, device="cpu")[: dim // 2].double() / dim))
    freqs = torch.outer(torch.arange(end, device="cpu"), freqs)      # explicit cpu: models are built under device("meta")
    return torch.polar(torch.ones_like(freqs), freqs)


def precompute_freqs_cis_3d(dim, end=1024, theta=10000.0):
    """(frame, height, width) tables; head_dim split d-2*(d//3), d//3, d//3 (reference :74-79)."""
    return (precompute_freqs_cis(dim - 2 * (dim // 3), end, theta),
            precompute_freqs_cis(dim // 3, end, theta),
            precompute_freqs_cis(dim // 3, end, theta))


# The bias-GEMMs of the 30 blocks (hipBLASLt behind PyTorch, as north_star leaves them).  Module-level seams so that
# bench.py can put HIP events around exactly these launches (its hipBLASLt roofline entry); looked up at call time.
def gemm_bias(x, weight, bias):
    return F.linear(x, weight, bias)


def gemm_bias_gelu(x, weight, bias):
    """ffn.0 with GELU(tanh) applied to the fp32 accumulator in the hipBLASLt epilogue; x (1, n, K)."""
    return torch._addmm_activation(bias, x[0], weight.t(), use_gelu=True).unsqueeze(0)


def gemm_bias_tuned(x, weight, bias):
    return tuning.linear(x, weight, bias)      # table: better hipBLASLt solution at the multi-GPU shard sizes


# The DiT Linears that run on this repo's own persistent MFMA kernel (fg_gemm_epilogue_bf16 / fg_gemm_fp8_bf16, csrc/gen_gemm_p.py) —
# also seams for bench.py.  FAIRYGEN_GEMM = "all" (default): every Linear of the 30 blocks — qkv, the self-attention o with the
# gate_msa add in its store, cross-attention q and o (+ residual), ffn.0 with GELU(tanh) in its epilogue, ffn.2 with the gate_mlp add —
# so a denoise step launches no library GEMM inside the blocks; "fused": qkv and ffn.0 (+ GELU in the hipBLASLt epilogue) stay on the
# library (round 2's default); "fused-ffn2": ffn.2 as well; "lib": everything on hipBLASLt.  Measurements: DESIGN.md §5.
GEMM_BACKEND = os.environ.get("FAIRYGEN_GEMM", "all")
GEMM_MIN_TILES = 64      # round 2 asked for two rounds of the CUs (512 tiles); see own_gemm_ok
FP8_FOLD = os.environ.get("FAIRYGEN_FP8_FOLD", "1") != "0"      # fp8 mode: norm kernels emit (e4m3 rows, scales) directly
FP8_GEMM = os.environ.get("FAIRYGEN_FP8_GEMM", "own")           # fp8 mode: "own" = fg_gemm_fp8_bf16, "lib" = torch._scaled_mm (hipBLASLt)


def own_gemm_ok(rows, n, k):
    """Shapes the persistent kernel takes: 256-column tiles, 128-byte k-steps in pairs, and enough tiles to be worth a 256-workgroup
    launch.  Round 2 asked for two rounds of the CUs (512 tiles); measured at the 8- / 4- / 2-rank shard sizes (3 410 / 6 820 / 13 640
    rows) the kernel is within 5 % of the library on every shape with one round or less, 1.6x faster on ffn.2 (K = 14 336:
    1 154 vs 739 TFLOP/s at 3 410 rows), and the fused residual store replaces a separate pass."""
    return GEMM_BACKEND != "lib" and n % 256 == 0 and k % 128 == 0 and ((rows + 255) // 256) * (n // 256) >= GEMM_MIN_TILES


def gemm_bias_own(x, weight, bias):
    return hip.gemm_epilogue(x, weight, bias)


def gemm_bias_gelu_own(x, weight, bias):
    """ffn.0 + nn.GELU(approximate='tanh') (models/wan_video_dit.py:208): GELU on the bf16-rounded Linear output in the GEMM's store."""
    return hip.gemm_epilogue(x, weight, bias, act="gelu_tanh")


def gemm_fp8_own(xq, scale_a, w8, bias):
    """fp8_linear's matmul on the e4m3 form of the persistent kernel: (rows, K) e4m3 x (N, K) e4m3 -> (1, rows, N) bf16."""
    return hip.gemm_fp8(xq, scale_a, w8, bias, lead_shape=(1, xq.shape[0]))


def gemm_residual(x, a, weight, bias, mod=None, gate_idx=None):
    """x (the residual stream, contiguous, updated in place) += gate * Linear(a); gate = vector gate_idx of mod, or 1."""
    return hip.gemm_epilogue(a, weight, bias, out=x, residual=True, mod=mod, gate_idx=gate_idx)


class RMSNorm(nn.Module):
    def __init__(self, dim, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))


class AttentionModule(nn.Module):
    """Plug point (iii) of the reference (:113-120): "b s (n d)" bf16 in and out, on the MFMA kernel."""

    def __init__(self, num_heads):
        super().__init__()
        self.num_heads = num_heads

    def forward(self, q, k, v, scale=None):
        return hip.attention(q, k, v, self.num_heads, scale=scale)


class SelfAttention(nn.Module):
    def __init__(self, dim, num_heads, eps=1e-6):
        super().__init__()
        self.dim, self.num_heads, self.head_dim = dim, num_heads, dim // num_heads
        self.q, self.k, self.v, self.o = (nn.Linear(dim, dim) for _ in range(4))
        self.norm_q, self.norm_k = RMSNorm(dim, eps=eps), RMSNorm(dim, eps=eps)
        self.attn = AttentionModule(num_heads)


class CrossAttention(nn.Module):
    def __init__(self, dim, num_heads, eps=1e-6, has_image_input=False):
        super().__init__()
        if has_image_input:
            raise NotImplementedError("has_image_input=True (Wan2.1 I2V CLIP branch) is outside the TI2V-5B hot path")
        self.dim, self.num_heads, self.head_dim = dim, num_heads, dim // num_heads
        self.q, self.k, self.v, self.o = (nn.Linear(dim, dim) for _ in range(4))
        self.norm_q, self.norm_k = RMSNorm(dim, eps=eps), RMSNorm(dim, eps=eps)
        self.has_image_input = has_image_input
        self.attn = AttentionModule(num_heads)


class DiTBlock(nn.Module):
    def __init__(self, has_image_input, dim, num_heads, ffn_dim, eps=1e-6):
        super().__init__()
        self.dim, self.num_heads, self.ffn_dim, self.eps = dim, num_heads, ffn_dim, eps
        self.self_attn = SelfAttention(dim, num_heads, eps)
        self.cross_attn = CrossAttention(dim, num_heads, eps, has_image_input=has_image_input)
        self.norm1 = nn.LayerNorm(dim, eps=eps, elementwise_affine=False)
        self.norm2 = nn.LayerNorm(dim, eps=eps, elementwise_affine=False)
        self.norm3 = nn.LayerNorm(dim, eps=eps)
        self.ffn = nn.Sequential(nn.Linear(dim, ffn_dim), nn.GELU(approximate="tanh"), nn.Linear(ffn_dim, dim))
        self.modulation = nn.Parameter(torch.randn(1, 6, dim) / dim ** 0.5)
        self._fused = None
        self._fp8 = None

    def fp8_weights(self, dtype):
        """The block's Linear weights cast to fp8 once (the reference casts on every call, core/vram/layers.py:342 — same
        values): [Wq;Wk;Wv], Wo, cross Wq, cross [Wk;Wv], cross Wo, ffn.0, ffn.2.  Rebuilt after load / LoRA fuse."""
        if self._fp8 is None or self._fp8[0] != dtype:
            wqkv, _, wkv_c, _ = self.fused_weights()
            sa, ca = self.self_attn, self.cross_attn
            ws = (wqkv, sa.o.weight, ca.q.weight, wkv_c, ca.o.weight, self.ffn[0].weight, self.ffn[2].weight)
            self._fp8 = (dtype, tuple(w.to(dtype).contiguous() for w in ws))
        return self._fp8[1]

    def fused_weights(self):
        """[Wq;Wk;Wv] and [Wk;Wv] (cross) concatenated once, so the three projections of a token tensor are
        ONE hipBLASLt GEMM.  Rebuilt after load_state_dict / LoRA fuse (see WanModel.invalidate_fused)."""
        if self._fused is None:
            sa, ca = self.self_attn, self.cross_attn
            self._fused = (
                torch.cat([sa.q.weight, sa.k.weight, sa.v.weight], 0).contiguous(),
                torch.cat([sa.q.bias, sa.k.bias, sa.v.bias], 0).contiguous(),
                torch.cat([ca.k.weight, ca.v.weight], 0).contiguous(),
                torch.cat([ca.k.bias, ca.v.bias], 0).contiguous(),
            )
        return self._fused


class Head(nn.Module):
    def __init__(self, dim, out_dim, patch_size, eps):
        super().__init__()
        self.dim, self.patch_size, self.eps = dim, patch_size, eps
        self.norm = nn.LayerNorm(dim, eps=eps, elementwise_affine=False)
        self.head = nn.Linear(dim, out_dim * math.prod(patch_size))
        self.modulation = nn.Parameter(torch.randn(1, 2, dim) / dim ** 0.5)


class WanModel(nn.Module):
    def __init__(self, dim, in_dim, ffn_dim, out_dim, text_dim, freq_dim, eps, patch_size, num_heads, num_layers,
                 has_image_input, has_image_pos_emb=False, has_ref_conv=False, add_control_adapter=False,
                 in_dim_control_adapter=24, seperated_timestep=False, require_vae_embedding=True,
                 require_clip_embedding=True, fuse_vae_embedding_in_latents=False):
        super().__init__()
        if has_image_input or has_ref_conv or add_control_adapter:
            raise NotImplementedError("CLIP image branch / ref_conv / camera adapter belong to other Wan variants "
                                      "(out of scope: SURVEY.md §2.1 rows 12)")
        self.dim, self.in_dim, self.out_dim, self.freq_dim = dim, in_dim, out_dim, freq_dim
        self.num_heads, self.eps = num_heads, eps
        self.has_image_input = has_image_input
        self.patch_size = tuple(patch_size)
        self.seperated_timestep = seperated_timestep
        self.require_vae_embedding = require_vae_embedding
        self.require_clip_embedding = require_clip_embedding
        self.fuse_vae_embedding_in_latents = fuse_vae_embedding_in_latents
        self.has_image_pos_emb, self.has_ref_conv, self.control_adapter = has_image_pos_emb, has_ref_conv, None

        self.patch_embedding = nn.Conv3d(in_dim, dim, kernel_size=self.patch_size, stride=self.patch_size)
        self.text_embedding = nn.Sequential(nn.Linear(text_dim, dim), nn.GELU(approximate="tanh"), nn.Linear(dim, dim))
        self.time_embedding = nn.Sequential(nn.Linear(freq_dim, dim), nn.SiLU(), nn.Linear(dim, dim))
        self.time_projection = nn.Sequential(nn.SiLU(), nn.Linear(dim, dim * 6))
        self.blocks = nn.ModuleList([DiTBlock(has_image_input, dim, num_heads, ffn_dim, eps) for _ in range(num_layers)])
        self.head = Head(dim, out_dim, self.patch_size, eps)
        self.freqs = precompute_freqs_cis_3d(dim // num_heads)
        self._rope_cache = {}
        # "f32" (default): RoPE as two fp32 FMAs on a table rounded once from the reference's complex128 one (the fp64
        # rotation costs 43 us of VALU per call on a 77 us HBM-bound kernel; < 0.2 % of the outputs move, by 1 bf16 ulp).
        # "f64": the reference's arithmetic exactly (rope_apply upcasts to complex128).
        self.rope_mode = "f32"
        # True (with rope_mode "f32" and the stock AttentionModule): self-attention's 1/sqrt(d) * log2(e) = 2^-3 * 1.0201 is split — the
        # 1.0201 goes into q's fp32 RoPE table (q is rounded to bf16 once, as before, from a value 2 % larger), attention is called with
        # scale' = 2^-3 / log2(e), for which the kernel's pre-multiplied form (64 VALU operations fewer per tile and wave, -4 % time) is
        # exact.  False: q as the reference rounds it, the kernel's plain form.
        self.fold_attn_scale = True
        # True: GELU(tanh) applied to the fp32 accumulator in the hipBLASLt epilogue of ffn.0 (one pass less over the
        # (n, ffn) tensor, -1.9 % per forward; verified to be the tanh form, tools/gelu_epilogue_check.py; <= 1 bf16 ulp
        # from the reference's "round, then GELU" order).  False: GEMM, then fg_act_bf16 on the rounded output.
        self.gelu_epilogue = True
        # hot-loaded (unfused) LoRA adapters: module name -> [(alpha*A (r,in), B (out,r)), ...]; see add_hot_lora
        self.hot_loras = {}
        # fp8 Linear mode of the blocks (None = bf16 GEMMs); see enable_fp8_linear
        self.fp8_dtype = None
        self._ones = {}

    # ------------------------------------------------------------------ load-time hooks
    def invalidate_fused(self):
        for blk in self.blocks:
            blk._fused = None
            blk._fp8 = None

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self.invalidate_fused()
        return out

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self.invalidate_fused()
        self._rope_cache = {}
        return out

    # ------------------------------------------------------------------ fp8 Linear mode (core/vram/layers.py:312,321-357)
    def enable_fp8_linear(self, dtype=torch.float8_e4m3fn):
        """Run the ten Linears of every DiT block like AutoWrappedLinear.fp8_linear when its computation_dtype is fp8:
        per-row dynamic activation scale (fg_fp8_quant_rows_bf16), weights cast to e4m3 with unit scale, bf16 bias,
        the row-scaled e4m3 matmul with bf16 output on fg_gemm_fp8_bf16 (or torch._scaled_mm: FAIRYGEN_FP8_GEMM=lib).  dtype None switches back to bf16 GEMMs.  The small
        embedding / head Linears (< 1 % of the FLOPs) stay bf16.  gfx950's fp8 GEMM takes the OCP format only, so
        torch.float8_e4m3fnuz (the MI300-era flavour the reference also accepts) is refused."""
        if dtype is not None and dtype != torch.float8_e4m3fn:
            raise NotImplementedError(f"{dtype}: the gfx950 fp8 GEMM (hipBLASLt behind torch._scaled_mm) takes "
                                      "torch.float8_e4m3fn (OCP e4m3) only")
        self.fp8_dtype = dtype
        self.invalidate_fused()
        return self

    def _scaled_linear(self, xq, scale_a, w8, bias):
        """fp8_linear's matmul (:347-354: torch._scaled_mm with row-wise scale_a, unit scale_b, bf16 bias, bf16 out): on this repo's e4m3
        MFMA kernel (fg_gemm_fp8_bf16), or — FAIRYGEN_FP8_GEMM=lib, and for shapes the kernel does not take — the library call itself."""
        n = w8.shape[0]
        if FP8_GEMM == "own" and n % 256 == 0 and xq.shape[1] % 256 == 0:
            return gemm_fp8_own(xq, scale_a, w8, bias)
        key = (n, xq.device)
        if key not in self._ones:
            self._ones[key] = torch.ones((1, n), dtype=torch.float32, device=xq.device)
        return torch._scaled_mm(xq, w8.T, scale_a=scale_a, scale_b=self._ones[key], bias=bias,
                                out_dtype=torch.bfloat16).unsqueeze(0)

    # ------------------------------------------------------------------ hot-loaded LoRA (core/vram/layers.py:417-436)
    def add_hot_lora(self, name, lora_a, lora_b):
        """Attach an unfused adapter to Linear `name` ("blocks.3.self_attn.q", ...): its output becomes
        linear(x) + x @ A^T @ B^T, evaluated left to right in the pipeline dtype like AutoWrappedLinear.lora_forward
        (`lora_a` already carries alpha, base_pipeline.py:258).  Adapters stack; clear_hot_loras() removes them all."""
        mod = dict(self.named_modules()).get(name)
        if not isinstance(mod, nn.Linear):
            raise KeyError(f"{name} is not a Linear of this model")
        if lora_a.shape[1] != mod.in_features or lora_b.shape[0] != mod.out_features or lora_a.shape[0] != lora_b.shape[1]:
            raise ValueError(f"LoRA shapes {tuple(lora_a.shape)}, {tuple(lora_b.shape)} do not fit {name} "
                             f"({mod.in_features} -> {mod.out_features})")
        w = mod.weight
        self.hot_loras.setdefault(name, []).append((lora_a.to(device=w.device, dtype=w.dtype).contiguous(),
                                                    lora_b.to(device=w.device, dtype=w.dtype).contiguous()))

    def clear_hot_loras(self):
        n = len(self.hot_loras)
        self.hot_loras = {}
        return n

    def _hot(self, name, x, out):
        """out (+)= sum of the adapters of `name` applied to x; `out` may be a column slice of a fused projection."""
        for a, b in self.hot_loras.get(name, ()):
            out += (x @ a.T) @ b.T
        return out

    # ------------------------------------------------------------------ host-side tables
    def rope_tables(self, f, h, w, device):
        """(cos, sin) of the per-token complex table, tokens frame-major (pipelines/wan_video.py:1271-1275): two fp64
        (N, 64) tensors in rope_mode "f64"; in rope_mode "f32" two interleaved fp32 (N, 64, 2) tensors {cos, sin}: the table for k and
        the table for q (the same one, or multiplied by the folded softmax-scale factor: attn_scale())."""
        fold = self.attn_scale()[1]
        key = (f, h, w, str(device), self.rope_mode, fold)
        if key not in self._rope_cache:
            tab = torch.cat([
                self.freqs[0][:f].view(f, 1, 1, -1).expand(f, h, w, -1),
                self.freqs[1][:h].view(1, h, 1, -1).expand(f, h, w, -1),
                self.freqs[2][:w].view(1, 1, w, -1).expand(f, h, w, -1),
            ], dim=-1).reshape(f * h * w, -1)
            if self.rope_mode == "f32":
                cs = torch.stack([tab.real, tab.imag], dim=-1)
                cs_k = cs.to(torch.float32).contiguous().to(device)
                self._rope_cache = {key: (cs_k, cs_k if fold == 1.0 else (cs * fold).to(torch.float32).contiguous().to(device))}
            else:
                self._rope_cache = {key: (tab.real.contiguous().to(device), tab.imag.contiguous().to(device))}
        return self._rope_cache[key]

    def attn_scale(self):
        """(scale passed to self-attention or None for 1/sqrt(d), factor folded into q's RoPE table)."""
        if self.fold_attn_scale and self.rope_mode == "f32" and all(type(b.self_attn.attn) is AttentionModule for b in self.blocks):
            return hip.pow2_softmax_scale(self.dim // self.num_heads)
        return None, 1.0

    def patchify(self, x):
        """Conv3d with kernel == stride == (1,p,p) is a GEMM over unfolded patches; returns frame-major
        tokens (b, f*h*w, dim) and the grid (reference :338-344 + pipelines/wan_video.py:1260-1261)."""
        b, c, t, hh, ww = x.shape
        pt, ph, pw = self.patch_size
        f, h, w = t // pt, hh // ph, ww // pw
        cols = x.view(b, c, f, pt, h, ph, w, pw).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(b, f * h * w, c * pt * ph * pw)
        wmat = self.patch_embedding.weight.view(self.dim, -1)
        return F.linear(cols, wmat, self.patch_embedding.bias), (f, h, w)

    def unpatchify(self, x, grid_size):
        """'b (f h w) (x y z c) -> b c (f x) (h y) (w z)' (reference :346-351)."""
        f, h, w = grid_size
        px, py, pz = self.patch_size
        b = x.shape[0]
        x = x.view(b, f, h, w, px, py, pz, self.out_dim).permute(0, 7, 1, 4, 2, 5, 3, 6)
        return x.reshape(b, self.out_dim, f * px, h * py, w * pz)

    def _self_attention_steps(self, i, blk, mod, x, h, cos, sin, shard, shard_total, own, lin):
        """Block i's self-attention half (reference :139-146, :225-226): h = modulate(norm1(x)) in; returns the residual stream after
        x += gate_msa * o(attn(...)) and h = norm3(x).  Yields right after each exchange of a token-sharded run has been started.
        lin: the fp8 Linear closure of the block (None in bf16 mode)."""
        c, nh, eps = self.dim, self.num_heads, self.eps
        fp8 = self.fp8_dtype if lin is not None else None
        hot = bool(self.hot_loras)
        sharded = shard is not None and shard.active
        wqkv, bqkv, _, _ = blk.fused_weights()
        sa = blk.self_attn
        if fp8 is not None:
            qkv = lin(h, 0, bqkv)
        elif GEMM_BACKEND == "all" and own_gemm_ok(h.shape[1], 3 * c, c):      # a plain Linear: hot adapters add to its output below
            qkv = gemm_bias_own(h, wqkv, bqkv)
        else:
            qkv = gemm_bias(h, wqkv, bqkv)
        if hot:
            for j, nm in enumerate(("q", "k", "v")):
                self._hot(f"blocks.{i}.self_attn.{nm}", h, qkv[..., j * c:(j + 1) * c])
        v = qkv[..., 2 * c:]
        # rope tables per operand: fp64 (cos, sin) for both, or the fp32 interleaved tables of k and q (rope_tables)
        rk, rq = ((cos, None), (sin, None)) if cos.dtype == torch.float32 else ((cos, sin), (cos, sin))
        scale = self.attn_scale()[0]      # None (1/sqrt(d)), or the power-of-two form that goes with the pre-multiplied q table
        if not sharded:
            k = hip.rmsnorm_rope(qkv[..., c:2 * c], sa.norm_k.weight, nh, eps, *rk)
            q = hip.rmsnorm_rope(qkv[..., :c], sa.norm_q.weight, nh, eps, *rq)
            a = sa.attn(q, k, v) if scale is None else sa.attn(q, k, v, scale=scale)
        elif shard.attn_mode == "ulysses":
            # token shard -> head shard (all N tokens of 24/P heads), attention, head shard -> token shard.  The
            # norm+RoPE kernels and one strided copy write q | k | v straight into the all-to-all send buffer.
            n_loc, p_, size = x.shape[1], shard.world_size, shard.chunk(shard_total)
            g = c // nh * shard.heads_local(nh)
            send = shard.ulysses_send_buffer(shard_total, c, qkv, n_loc)           # (P, chunk, 3, g)
            flat, layout = send.view(-1), (g, size * 3 * g, 3 * g)
            hip.rmsnorm_rope(qkv[..., :c], sa.norm_q.weight, nh, eps, *rq, grouped=(flat, *layout))
            hip.rmsnorm_rope(qkv[..., c:2 * c], sa.norm_k.weight, nh, eps, *rk, grouped=(flat[g:], *layout))
            hip.copy_groups(qkv.view(-1)[2 * c:], g, 3 * c, flat[2 * g:], size * 3 * g, 3 * g, p_, n_loc, g)
            pending = shard.ulysses_exchange_async(send, shard_total)
            yield i
            qg, kg, vg = pending.wait()
            o_full = shard.ulysses_out_buffer(shard_total, g, qkv)
            hip.attention(qg, kg, vg, shard.heads_local(nh), out=o_full[:shard_total].unsqueeze(0), scale=scale)
            pending = shard.ulysses_out_async(o_full, shard_total, n_loc)
            yield i
            a = torch.empty((1, n_loc, c), dtype=qkv.dtype, device=qkv.device)
            hip.copy_groups(pending.wait_blocks().view(-1), size * g, g, a.view(-1), g, c, p_, n_loc, g)
        else:
            k = hip.rmsnorm_rope(qkv[..., c:2 * c], sa.norm_k.weight, nh, eps, *rk)
            pending = shard.all_gather_kv_async(k, v, shard_total)
            q = hip.rmsnorm_rope(qkv[..., :c], sa.norm_q.weight, nh, eps, *rq)
            yield i
            k, v = pending.wait()
            a = sa.attn(q, k, v) if scale is None else sa.attn(q, k, v, scale=scale)
        # x += gate_msa*y ; h = norm3(x)  (reference :225-226)
        if own and own_gemm_ok(a.shape[1], c, c):      # the gated add happens in the GEMM's store
            x = gemm_residual(x, a, sa.o.weight, sa.o.bias, mod, 2)
            h = hip.ln_affine(x, blk.norm3.weight, blk.norm3.bias, eps)
        else:
            y = gemm_bias(a, sa.o.weight, sa.o.bias) if fp8 is None else lin(a, 1, sa.o.bias)
            if hot:
                self._hot(f"blocks.{i}.self_attn.o", a, y)
            if fp8 is not None and not hot and FP8_FOLD:
                x, h = hip.residual_ln_affine_fp8(x, y, blk.norm3.weight, blk.norm3.bias, eps, mod, 2, x_out=x)
            else:
                x, h = hip.residual_ln_affine(x, y, blk.norm3.weight, blk.norm3.bias, eps, mod, 2, x_out=x)
        return x, h

    # ------------------------------------------------------------------ the 30-block token forward
    def forward_tokens(self, x, context, mod_rows_t, t_rows, first_rows, rope, shard=None, shard_total=None, tea_cache=None,
                       skip_blocks=False, cfg_prefix=None, kv_cache=None):
        """Run forward_tokens_steps to completion (single branch)."""
        gen = self.forward_tokens_steps(x, context, mod_rows_t, t_rows, first_rows, rope, shard, shard_total, tea_cache, skip_blocks,
                                        cfg_prefix, kv_cache)
        while True:
            try:
                next(gen)
            except StopIteration as done:
                return done.value

    def forward_tokens_steps(self, x, context, mod_rows_t, t_rows, first_rows, rope, shard=None, shard_total=None,
                             tea_cache=None, skip_blocks=False, cfg_prefix=None, kv_cache=None):
        """Generator form of the 30-block forward: yields right after each of a block's exchanges has been STARTED
        (token-sharded runs only: the K/V all-gather, or the two Ulysses all-to-alls), so a driver can interleave two
        independent forwards (the CFG branches) and let one branch's compute hide the other's xGMI traffic.
        Returns (StopIteration.value) the head output.

        x (1,n,dim) local tokens; context (1,L,dim) embedded text; t_rows (R,dim) distinct time embeddings
        (R = 1 or 2), mod_rows_t (R,6,dim) their projections; tokens < first_rows use row 0.
        rope = (cos, sin) for the LOCAL tokens.  shard: optional fairygen_amd.sequence_parallel.TokenShard —
        its attn_mode picks the exchange around self-attention (shard_total = N, all ranks' tokens).
        tea_cache / skip_blocks: the wan_video.TeaCache of this CFG branch and its verdict for this step — a skipped step
        re-applies the cached residual instead of running the 30 blocks, a computed one stores the new residual
        (pipelines/wan_video.py:1297-1300,1316-1317,1375-1376).
        cfg_prefix: a dict shared by the forwards of ONE denoise step that differ only in `context` (the CFG branches,
        pipelines/wan_video.py:296-301).  Nothing before block 0's cross-attention sees the context, so the residual stream after
        block 0's self-attention (norm1 + modulate, qkv, RMSNorm + RoPE, attention and its exchanges, o, gate) is the same tensor in
        both: the first forward to get there leaves a copy, the other one takes it instead of computing it (bit-identical: the
        kernels are deterministic).  A forward that arrives while the other is still inside that self-attention (lockstep
        interleave) yields until the copy is there.
        kv_cache: a dict that lives as long as `context` and the weights stay what they are (one denoise loop): block i's
        cross-attention keys (after norm_k) and values depend on nothing else, so they are computed at the first step and read
        back at the others (reference :172-177 recomputes them every step)."""
        c, nh, eps = self.dim, self.num_heads, self.eps
        cos, sin = rope
        x = x.contiguous()
        blocks = list(self.blocks)
        if skip_blocks:
            blocks, x = [], tea_cache.update(x)
        sharded = shard is not None and shard.active
        hot = bool(self.hot_loras)
        fp8 = self.fp8_dtype
        ctx8 = hip.fp8_quant_rows(context) if fp8 is not None else None      # the text context is the same for all blocks
        mods = [hip.ModTable((blk.modulation.to(mod_rows_t.dtype) + mod_rows_t).contiguous(), first_rows) for blk in blocks]
        # fp8 mode: the four norms of a block feed fp8 Linears only, so they hand over (e4m3 rows, row scales) directly;
        # a hot-loaded adapter needs the bf16 row too: then the separate quantisation stays
        fold8 = fp8 is not None and not hot and FP8_FOLD
        h = (hip.ln_modulate_fp8 if fold8 else hip.ln_modulate)(x, mods[0], 0, 1, eps) if blocks else None
        for i, blk in enumerate(blocks):
            mod = mods[i]
            wqkv, bqkv, wkv_c, bkv_c = blk.fused_weights()
            sa, ca = blk.self_attn, blk.cross_attn
            if fp8 is not None:
                w8 = blk.fp8_weights(fp8)
                # t: a bf16 tensor (quantised here) or the (fp8 rows, scales) pair a fused norm kernel already produced
                lin = lambda t, j, bias, act=None: self._scaled_linear(*(t if isinstance(t, tuple) else hip.fp8_quant_rows(t, act)), w8[j], bias)      # noqa: E731
            own = fp8 is None and not hot and mod.mod_rows in (1, 2)
            if cfg_prefix is not None and i == 0 and "owner" in cfg_prefix:
                if cfg_prefix.get("taken"):
                    raise RuntimeError("cfg_prefix is shared by exactly two forwards of one step (one computes block 0's self-attention half, "
                                       "one takes it): a third consumer of the same dict")
                spins = 0
                while "x_sa" not in cfg_prefix:      # the other branch is inside block 0's self-attention: let it run
                    spins += 1
                    if spins > 100000:
                        raise RuntimeError("cfg_prefix: the forward that owns block 0's self-attention half never published it (the two "
                                           "forwards of a step must be advanced in turns, or run one after the other)")
                    yield i
                x = cfg_prefix.pop("x_sa")
                cfg_prefix["taken"] = True
                h = hip.ln_affine(x, blk.norm3.weight, blk.norm3.bias, eps)
                reuse = True
            else:
                reuse = False
                if cfg_prefix is not None and i == 0:
                    cfg_prefix["owner"] = True
            # --- self attention (reference :139-146)
            if not reuse:
                x, h = yield from self._self_attention_steps(i, blk, mod, x, h, cos, sin, shard, shard_total, own, lin if fp8 is not None else None)
                if cfg_prefix is not None and i == 0:
                    cfg_prefix["x_sa"] = x.clone()
            # --- cross attention (reference :170-185)
            if fp8 is None:
                qc = gemm_bias_own(h, ca.q.weight, ca.q.bias) if own and own_gemm_ok(h.shape[1], c, c) else gemm_bias(h, ca.q.weight, ca.q.bias)
            else:
                qc = lin(h, 2, ca.q.bias)
            if hot:
                self._hot(f"blocks.{i}.cross_attn.q", h, qc)
            qc = hip.rmsnorm_rope(qc, ca.norm_q.weight, nh, eps)
            if kv_cache is not None and i in kv_cache:
                kc, vc = kv_cache[i]
            else:
                kvc = gemm_bias(context, wkv_c, bkv_c) if fp8 is None else self._scaled_linear(*ctx8, w8[3], bkv_c)
                if hot:
                    self._hot(f"blocks.{i}.cross_attn.k", context, kvc[..., :c])
                    self._hot(f"blocks.{i}.cross_attn.v", context, kvc[..., c:])
                kc, vc = hip.rmsnorm_rope(kvc[..., :c], ca.norm_k.weight, nh, eps), kvc[..., c:]
                if kv_cache is not None:
                    kv_cache[i] = (kc, vc)
            ac = ca.attn(qc, kc, vc)
            # x += y ; h = modulate(norm2(x))  (reference :226-227)
            if own and own_gemm_ok(ac.shape[1], c, c):
                x = gemm_residual(x, ac, ca.o.weight, ca.o.bias)
                h = hip.ln_modulate(x, mod, 3, 4, eps)
            else:
                y = gemm_bias(ac, ca.o.weight, ca.o.bias) if fp8 is None else lin(ac, 4, ca.o.bias)
                if hot:
                    self._hot(f"blocks.{i}.cross_attn.o", ac, y)
                if fold8:
                    x, h = hip.residual_ln_modulate_fp8(x, y, mod, None, 3, 4, eps, x_out=x)
                else:
                    x, h = hip.residual_ln_modulate(x, y, mod, None, 3, 4, eps, x_out=x)
            # --- ffn (reference :208-209,228)
            if fp8 is not None:      # ffn.0 -> bf16, GELU(tanh) fused into the quantisation of ffn.2's input
                pre = lin(h, 5, blk.ffn[0].bias)
                if hot:
                    self._hot(f"blocks.{i}.ffn.0", h, pre)
                y = lin(pre, 6, blk.ffn[2].bias, "gelu_tanh")
                f = None
            elif hot and f"blocks.{i}.ffn.0" in self.hot_loras:      # the adapter adds to the pre-activation: no epilogue fusion
                f = hip.activation(self._hot(f"blocks.{i}.ffn.0", h, gemm_bias(h, blk.ffn[0].weight, blk.ffn[0].bias)), "gelu_tanh")
            elif GEMM_BACKEND == "all" and own_gemm_ok(h.shape[1], blk.ffn[0].weight.shape[0], c):      # GELU(tanh) in the own GEMM's store
                f = gemm_bias_gelu_own(h, blk.ffn[0].weight, blk.ffn[0].bias)
            elif self.gelu_epilogue:      # GELU(tanh) in the hipBLASLt epilogue: one pass less over the (n, ffn) tensor
                f = gemm_bias_gelu(h, blk.ffn[0].weight, blk.ffn[0].bias)
            else:
                f = hip.activation(gemm_bias(h, blk.ffn[0].weight, blk.ffn[0].bias), "gelu_tanh")
            if own and GEMM_BACKEND != "fused-ffn2" and own_gemm_ok(f.shape[1], c, f.shape[2]):      # x += gate_mlp * ffn.2(f) in the store
                x = gemm_residual(x, f, blk.ffn[2].weight, blk.ffn[2].bias, mod, 5)
                if i + 1 < len(blocks):
                    h = hip.ln_modulate(x, mods[i + 1], 0, 1, eps, out=h)
                continue
            if fp8 is None:
                y = gemm_bias_tuned(f, blk.ffn[2].weight, blk.ffn[2].bias)
            if hot and f"blocks.{i}.ffn.2" in self.hot_loras:
                if f is None:
                    f = hip.activation(pre.clone(), "gelu_tanh")
                self._hot(f"blocks.{i}.ffn.2", f, y)
            if i + 1 < len(blocks) and fold8:
                x, h = hip.residual_ln_modulate_fp8(x, y, mod, 5, 0, 1, eps, x_out=x, norm_mod=mods[i + 1])
            elif i + 1 < len(blocks):   # x += gate_mlp*y fused with the NEXT block's modulate(norm1(x))
                x, h = hip.residual_ln_modulate(x, y, mod, 5, 0, 1, eps, x_out=x, norm_out=h, norm_mod=mods[i + 1])
            else:
                x = hip.gate_residual(x, y, mod, 5, out=x)
        if tea_cache is not None and blocks:
            tea_cache.store(x)
        # --- head (reference :261-268): table (R,2,C) = modulation + t
        hm = hip.ModTable((self.head.modulation.to(t_rows.dtype) + t_rows.unsqueeze(1)).contiguous(), first_rows)
        h = hip.ln_modulate(x, hm, 0, 1, eps)
        return F.linear(h, self.head.head.weight, self.head.head.bias)

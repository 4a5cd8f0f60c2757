"""Oracle: Wan2.2-TI2V-5B DiT forward, functional CPU restatement (test infrastructure only).

Follows ``diffsynth/models/wan_video_dit.py`` and the TI2V branch of ``model_fn_wan_video``
(``diffsynth/pipelines/wan_video.py:1122-1388``).  Weights come in as a flat state dict using the
reference's parameter names, so a reference checkpoint, the synthetic checkpoints of
``fairygen_amd.synthetic`` and the golden fixtures all feed it unchanged.  Every op runs in the
dtype of the tensors handed in (bf16 in → the reference's bf16 CPU arithmetic, op for op).
"""
import math

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- embeddings / tables
def sinusoid_1d(dim, position):
    """wan_video_dit.py:67-71 — fp64 sinusoid [cos | sin], cast back to position.dtype."""
    pos = position.to(torch.float64)
    inv = torch.pow(10000.0, -torch.arange(dim // 2, dtype=torch.float64).div(dim // 2))
    ang = torch.outer(pos, inv)
    return torch.cat([ang.cos(), ang.sin()], dim=1).to(position.dtype)


def rope_axis_table(dim, end=1024, theta=10000.0):
    """wan_video_dit.py:82-88 — complex128 e^{i·pos·theta^(-2j/dim)}, shape (end, dim//2)."""
    expo = torch.arange(0, dim, 2)[: dim // 2].double() / dim
    ang = torch.outer(torch.arange(end), 1.0 / theta ** expo)
    return torch.polar(torch.ones_like(ang), ang)


def rope_table_3d(head_dim, f, h, w):
    """wan_video_dit.py:74-79 + wan_video.py:1271-1275 — per-token table (f*h*w, 1, head_dim//2).

    head_dim is split frame/height/width as (d - 2*(d//3), d//3, d//3) reals; tokens are frame-major.
    """
    df, dh = head_dim - 2 * (head_dim // 3), head_dim // 3
    tf, th, tw = rope_axis_table(df), rope_axis_table(dh), rope_axis_table(dh)
    tab = torch.cat([
        tf[:f].view(f, 1, 1, -1).expand(f, h, w, -1),
        th[:h].view(1, h, 1, -1).expand(f, h, w, -1),
        tw[:w].view(1, 1, w, -1).expand(f, h, w, -1),
    ], dim=-1)
    return tab.reshape(f * h * w, 1, -1)


def rope_apply(x, table, num_heads):
    """wan_video_dit.py:91-96 — rotate adjacent pairs in fp64, cast back."""
    b, s, _ = x.shape
    xc = torch.view_as_complex(x.to(torch.float64).reshape(b, s, num_heads, -1, 2))
    return torch.view_as_real(xc * table).flatten(2).to(x.dtype)


# ----------------------------------------------------------------------------------------- norms
def rms_norm(x, weight, eps):
    """wan_video_dit.py:99-110 — fp32 normalise over the FULL last dim, cast, then * weight."""
    xf = x.float()
    y = xf * torch.rsqrt(xf.pow(2).mean(dim=-1, keepdim=True) + eps)
    return y.to(x.dtype) * weight


def layer_norm(x, eps, weight=None, bias=None):
    """nn.LayerNorm as used at wan_video_dit.py:205-207,257."""
    return F.layer_norm(x, (x.shape[-1],), weight, bias, eps)


def scaled_mm(a8, b8t, scale_a, scale_b, bias, out_dtype):
    """What torch._scaled_mm computes (row-wise scales): (a8 @ b8t) in fp32 * scale_a * scale_b + bias -> out_dtype.
    The CPU backend of this container only takes per-tensor scales, so the op is written out; with unit scales it is
    checked against the real CPU torch._scaled_mm in tests/test_oracle_golden.py."""
    return ((a8.float() @ b8t.float()) * scale_a * scale_b + bias.float()).to(out_dtype)


def fp8_linear(x, weight, bias, dtype=torch.float8_e4m3fn):
    """AutoWrappedLinear.fp8_linear, core/vram/layers.py:321-357: per-row dynamic activation scale (only ever scaling
    DOWN: clamp(min=1)), weights cast to fp8 with unit scale, bf16 bias, row-wise scaled matmul, result in x's dtype.
    Pinned on the shape the CPU backend of the build container accepts: torch._scaled_mm with (rows, 1) / (1, out) scales runs
    there only for one row x one output, and 64 such calls of the reference's own method (oracle/gen_fp8_linear_1x1.py, K = 3072 and
    14336, row maxima below / around / above fp8_max) are reproduced bit for bit (tests/test_oracle_golden.py).  Multi-row calls are
    rejected there (ordinary RuntimeError), so anything beyond one row per call — there is no cross-row arithmetic in the method —
    has no vector of its own."""
    origin_dtype, origin_shape = x.dtype, x.shape
    inp = x.reshape(-1, origin_shape[-1])
    x_max = torch.max(torch.abs(inp), dim=-1, keepdim=True).values
    fp8_max = 448.0 / 2.0 if dtype == torch.float8_e4m3fnuz else 448.0
    scale_a = torch.clamp(x_max / fp8_max, min=1.0).float()
    scale_b = torch.ones((weight.shape[0], 1))
    inp = (inp / (scale_a + 1e-8)).to(dtype)
    out = scaled_mm(inp, weight.to(dtype).T, scale_a, scale_b.T, bias.to(torch.bfloat16), origin_dtype)
    return out.reshape(origin_shape[:-1] + out.shape[-1:])


class Fp8Blocks(dict):
    """A state dict whose `blocks.*` Linears run as fp8_linear (the build's enable_fp8_linear scope)."""
    fp8_dtype = torch.float8_e4m3fn


def linear(sd, prefix, x):
    if getattr(sd, "fp8_dtype", None) is not None and prefix.startswith("blocks."):
        return fp8_linear(x, sd[prefix + ".weight"], sd[prefix + ".bias"], sd.fp8_dtype)
    return F.linear(x, sd[prefix + ".weight"], sd[prefix + ".bias"])


def attention(q, k, v, num_heads):
    """wan_video_dit.py:54-59 — the SDPA fallback: softmax(q k^T / sqrt(d)) v, no mask."""
    b, sq, c = q.shape
    d = c // num_heads
    qh = q.view(b, sq, num_heads, d).transpose(1, 2)
    kh = k.view(b, k.shape[1], num_heads, d).transpose(1, 2)
    vh = v.view(b, v.shape[1], num_heads, d).transpose(1, 2)
    o = F.scaled_dot_product_attention(qh, kh, vh)
    return o.transpose(1, 2).reshape(b, sq, c)


# --------------------------------------------------------------------------------------- modules
def self_attention(sd, p, x, table, num_heads, eps):
    """wan_video_dit.py:139-146."""
    q = rms_norm(linear(sd, p + ".q", x), sd[p + ".norm_q.weight"], eps)
    k = rms_norm(linear(sd, p + ".k", x), sd[p + ".norm_k.weight"], eps)
    v = linear(sd, p + ".v", x)
    q = rope_apply(q, table, num_heads)
    k = rope_apply(k, table, num_heads)
    return linear(sd, p + ".o", attention(q, k, v, num_heads))


def cross_attention(sd, p, x, ctx, num_heads, eps):
    """wan_video_dit.py:170-185, has_image_input=False branch."""
    q = rms_norm(linear(sd, p + ".q", x), sd[p + ".norm_q.weight"], eps)
    k = rms_norm(linear(sd, p + ".k", ctx), sd[p + ".norm_k.weight"], eps)
    v = linear(sd, p + ".v", ctx)
    return linear(sd, p + ".o", attention(q, k, v, num_heads))


def dit_block(sd, p, x, ctx, t_mod, table, num_heads, eps):
    """wan_video_dit.py:213-229 — AdaLN-Zero block; t_mod is (B,6,C) or per-token (B,N,6,C)."""
    per_token = t_mod.dim() == 4
    mod = sd[p + ".modulation"].to(dtype=t_mod.dtype) + t_mod
    parts = mod.chunk(6, dim=2 if per_token else 1)
    if per_token:
        parts = [u.squeeze(2) for u in parts]
    shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = parts
    h = layer_norm(x, eps) * (1 + scale_msa) + shift_msa
    x = x + gate_msa * self_attention(sd, p + ".self_attn", h, table, num_heads, eps)
    x = x + cross_attention(sd, p + ".cross_attn",
                            layer_norm(x, eps, sd[p + ".norm3.weight"], sd[p + ".norm3.bias"]),
                            ctx, num_heads, eps)
    h = layer_norm(x, eps) * (1 + scale_mlp) + shift_mlp
    h = linear(sd, p + ".ffn.2", F.gelu(linear(sd, p + ".ffn.0", h), approximate="tanh"))
    return x + gate_mlp * h


def head(sd, x, t, eps):
    """wan_video_dit.py:261-268 — t is (B,N,C) per token (TI2V) or (B,C)."""
    m = sd["head.modulation"].to(dtype=t.dtype)
    if t.dim() == 3:
        shift, scale = (m.unsqueeze(0) + t.unsqueeze(2)).chunk(2, dim=2)
        y = layer_norm(x, eps) * (1 + scale.squeeze(2)) + shift.squeeze(2)
    else:
        shift, scale = (m + t).chunk(2, dim=1)
        y = layer_norm(x, eps) * (1 + scale) + shift
    return linear(sd, "head.head", y)


def patchify_tokens(sd, latents, patch):
    """wan_video_dit.py:338-339 + wan_video.py:1260-1261 — Conv3d k=s=patch, then frame-major tokens."""
    x = F.conv3d(latents, sd["patch_embedding.weight"], sd["patch_embedding.bias"], stride=tuple(patch))
    f, h, w = x.shape[2:]
    return x.flatten(2).transpose(1, 2).contiguous(), (f, h, w)


def unpatchify(x, grid, patch, out_dim):
    """wan_video_dit.py:346-351 — 'b (f h w) (x y z c) -> b c (f x) (h y) (w z)'."""
    f, h, w = grid
    px, py, pz = patch
    b = x.shape[0]
    x = x.view(b, f, h, w, px, py, pz, out_dim).permute(0, 7, 1, 4, 2, 5, 3, 6)
    return x.reshape(b, out_dim, f * px, h * py, w * pz)


class TeaCache:
    """pipelines/wan_video.py:1016-1065 — skip the block stack when the accumulated, polynomial-rescaled relative L1
    change of the time modulation since the last computed step stays under a threshold; a skipped step re-applies the
    cached residual (x_after_blocks - x_before_blocks)."""

    COEFFICIENTS = {
        "Wan2.1-T2V-1.3B": [-5.21862437e+04, 9.23041404e+03, -5.28275948e+02, 1.36987616e+01, -4.99875664e-02],
        "Wan2.1-T2V-14B": [-3.03318725e+05, 4.90537029e+04, -2.65530556e+03, 5.87365115e+01, -3.15583525e-01],
        "Wan2.1-I2V-14B-480P": [2.57151496e+05, -3.54229917e+04, 1.40286849e+03, -1.35890334e+01, 1.32517977e-01],
        "Wan2.1-I2V-14B-720P": [8.10705460e+03, 2.13393892e+03, -3.72934672e+02, 1.66203073e+01, -4.17769401e-02],
    }

    def __init__(self, num_inference_steps, rel_l1_thresh, model_id):
        if model_id not in self.COEFFICIENTS:
            raise ValueError(f"{model_id} is not a supported TeaCache model id. Please choose a valid model id in "
                             f"({', '.join(self.COEFFICIENTS)}).")
        self.num_inference_steps, self.rel_l1_thresh = num_inference_steps, rel_l1_thresh
        self.coefficients = self.COEFFICIENTS[model_id]
        self.step, self.accumulated = 0, 0
        self.previous_modulated_input = self.previous_residual = self.previous_hidden_states = None

    def check(self, x, t_mod):
        """-> True when the blocks are to be SKIPPED this step (:1037-1058)."""
        import numpy as np
        if self.step == 0 or self.step == self.num_inference_steps - 1:
            should_calc, self.accumulated = True, 0
        else:
            rel = ((t_mod - self.previous_modulated_input).abs().mean() / self.previous_modulated_input.abs().mean()).cpu().item()
            self.accumulated += np.poly1d(self.coefficients)(rel)
            should_calc = not (self.accumulated < self.rel_l1_thresh)
            if should_calc:
                self.accumulated = 0
        self.previous_modulated_input = t_mod.clone()
        self.step = (self.step + 1) % self.num_inference_steps
        if should_calc:
            self.previous_hidden_states = x.clone()
        return not should_calc

    def store(self, x):
        self.previous_residual = x - self.previous_hidden_states
        self.previous_hidden_states = None

    def update(self, x):
        return x + self.previous_residual


def model_fn(sd, cfg, latents, timestep, context, fuse_vae_embedding_in_latents=True, num_blocks=None, tea_cache=None):
    """One DiT forward = model_fn_wan_video (wan_video.py:1217-1231,1236,1253-1275,1329-1362,1378,1387).

    cfg: dict with dim, num_heads, eps, patch_size, freq_dim, out_dim, seperated_timestep.
    """
    dim, nh, eps, patch = cfg["dim"], cfg["num_heads"], cfg["eps"], tuple(cfg["patch_size"])

    def time_embedding(e):
        return linear(sd, "time_embedding.2", F.silu(linear(sd, "time_embedding.0", e)))

    if cfg.get("seperated_timestep", False) and fuse_vae_embedding_in_latents:
        per_frame = latents.shape[3] * latents.shape[4] // 4
        ts = torch.cat([
            torch.zeros((1, per_frame), dtype=latents.dtype),
            torch.ones((latents.shape[2] - 1, per_frame), dtype=latents.dtype) * timestep,
        ]).flatten()
        t = time_embedding(sinusoid_1d(cfg["freq_dim"], ts).unsqueeze(0))
        t_mod = linear(sd, "time_projection.1", F.silu(t)).unflatten(2, (6, dim))
    else:
        t = time_embedding(sinusoid_1d(cfg["freq_dim"], timestep))
        t_mod = linear(sd, "time_projection.1", F.silu(t)).unflatten(1, (6, dim))

    ctx = linear(sd, "text_embedding.2", F.gelu(linear(sd, "text_embedding.0", context), approximate="tanh"))
    x, (f, h, w) = patchify_tokens(sd, latents, patch)
    table = rope_table_3d(dim // nh, f, h, w)
    nblocks = cfg["num_layers"] if num_blocks is None else num_blocks
    if tea_cache is not None and tea_cache.check(x, t_mod):        # wan_video.py:1297-1300,1316-1317
        x = tea_cache.update(x)
    else:
        for i in range(nblocks):
            x = dit_block(sd, f"blocks.{i}", x, ctx, t_mod, table, nh, eps)
        if tea_cache is not None:                                   # :1375-1376
            tea_cache.store(x)
    x = head(sd, x, t, eps)
    return unpatchify(x, (f, h, w), patch, cfg["out_dim"])


def temporal_tiler(model_fn_, latents, sliding_window_size, sliding_window_stride):
    """TemporalTiler_BCTHW.run, pipelines/wan_video.py:1069-1118: overlapping windows of latent frames, each an independent
    forward, blended with linear ramps of width (size - stride) in the data dtype.  `model_fn_(window) -> prediction`."""
    b, c, t_all, h, w = latents.shape
    value = torch.zeros((b, c, t_all, h, w), dtype=latents.dtype)
    weight = torch.zeros((1, 1, t_all, 1, 1), dtype=latents.dtype)
    border = sliding_window_size - sliding_window_stride
    for t in range(0, t_all, sliding_window_stride):
        if t - sliding_window_stride >= 0 and t - sliding_window_stride + sliding_window_size >= t_all:
            continue
        t_ = min(t + sliding_window_size, t_all)
        out = model_fn_(latents[:, :, t:t_])
        mask = torch.ones((t_ - t,))
        if border > 0:
            ramp = (torch.arange(border) + 0.5) / border
            if t != 0:
                mask[:border] = ramp
            if t_ != t_all:
                mask[-border:] = torch.flip(ramp, dims=(0,))
        mask = mask.view(1, 1, -1, 1, 1).to(latents.dtype)
        value[:, :, t:t_] += out * mask
        weight[:, :, t:t_] += mask
    value /= weight
    return value


def model_fn_sliding(sd, cfg, latents, timestep, context, sliding_window_size, sliding_window_stride, num_blocks=None):
    """model_fn_wan_video with sliding_window_size / stride (:1158-1182).  The window calls do not receive
    fuse_vae_embedding_in_latents (it is not in the reference's model_kwargs), so they run in the single-timestep mode."""
    return temporal_tiler(lambda win: model_fn(sd, cfg, win, timestep, context, False, num_blocks),
                          latents, sliding_window_size, sliding_window_stride)

"""Oracle: Wan2.2 VAE ("VAE38") chunked causal decode, functional CPU restatement (test infrastructure only).

Follows ``diffsynth/models/wan_video_vae.py``: ``VideoVAE38_.decode`` :1326-1351,
``Decoder3d_38.forward`` :889-940, ``Up_ResidualBlock`` :477-514, ``ResidualBlock`` :283-301,
``Resample.forward`` (upsample3d / upsample2d) :120-160, ``DupUp3D`` :417-439, ``AttentionBlock`` :321-342,
``CausalConv3d`` :44-52, ``RMS_norm`` :67-70, tiling ``WanVideoVAE.tiled_decode`` :1103-1152.
Weights arrive as a flat state dict with the reference's names (prefix ``model.``).
"""
import torch
import torch.nn.functional as F

CACHE_T = 2  # wan_video_vae.py:8

VAE38_MEAN = [
    -0.2289, -0.0052, -0.1323, -0.2339, -0.2799, 0.0174, 0.1838, 0.1557, -0.1382, 0.0542, 0.2813, 0.0891,
    0.1570, -0.0098, 0.0375, -0.1825, -0.2246, -0.1207, -0.0698, 0.5109, 0.2665, -0.2108, -0.2158, 0.2502,
    -0.2055, -0.0322, 0.1109, 0.1567, -0.0729, 0.0899, -0.2799, -0.1230, -0.0313, -0.1649, 0.0117, 0.0723,
    -0.2839, -0.2083, -0.0520, 0.3748, 0.0152, 0.1957, 0.1433, -0.2944, 0.3573, -0.0548, -0.1681, -0.0667,
]
VAE38_STD = [
    0.4765, 1.0364, 0.4514, 1.1677, 0.5313, 0.4990, 0.4818, 0.5013, 0.8158, 1.0344, 0.5894, 1.0901,
    0.6885, 0.6165, 0.8454, 0.4978, 0.5759, 0.3523, 0.7135, 0.6804, 0.5833, 1.4146, 0.8986, 0.5659,
    0.7069, 0.5338, 0.4889, 0.4917, 0.4069, 0.4999, 0.6866, 0.4093, 0.5709, 0.6065, 0.6415, 0.4944,
    0.5726, 1.2042, 0.5458, 1.6887, 0.3971, 1.0600, 0.3943, 0.5537, 0.5444, 0.4089, 0.7468, 0.7744,
]  # wan_video_vae.py:1359-1377


def causal_conv3d(sd, p, x, cache=None):
    """wan_video_vae.py:44-52 — left-pad time by 2*pad_t (or prepend the cache), zero-pad space."""
    w, b = sd[p + ".weight"], sd[p + ".bias"]
    kt, kh, kw = w.shape[2:]
    pt, ph, pw = kt - 1, kh // 2, kw // 2          # causal: 2*padding[0] == kt-1 for kt in {1,3}
    if cache is not None and pt > 0:
        x = torch.cat([cache, x], dim=2)
        pt -= cache.shape[2]
    x = F.pad(x, (pw, pw, ph, ph, pt, 0))
    return F.conv3d(x, w, b)


def rms_norm_c(sd, p, x):
    """wan_video_vae.py:67-70 — L2-normalise over channels, * sqrt(C) * gamma (bias is 0.)."""
    g = sd[p + ".gamma"]
    return F.normalize(x, dim=1) * (x.shape[1] ** 0.5) * g + 0.0


def _roll_cache(x, prev):
    """The cache-update idiom at wan_video_vae.py:288-297 / 892-902: last 2 input frames, topped up
    with the previous cache's last frame when the chunk has a single frame."""
    c = x[:, :, -CACHE_T:].clone()
    if c.shape[2] < 2 and prev is not None:
        c = torch.cat([prev[:, :, -1:].to(c.device), c], dim=2)
    return c


def residual_block(sd, p, x, cache, idx):
    """wan_video_vae.py:283-301 — [RMS, SiLU, conv3, RMS, SiLU, (dropout), conv3] + shortcut."""
    h = causal_conv3d(sd, p + ".shortcut", x) if (p + ".shortcut.weight") in sd else x
    for norm_i, conv_i in ((0, 2), (3, 6)):
        x = F.silu(rms_norm_c(sd, f"{p}.residual.{norm_i}", x))
        new_cache = _roll_cache(x, cache[idx[0]])
        x = causal_conv3d(sd, f"{p}.residual.{conv_i}", x, cache[idx[0]])
        cache[idx[0]] = new_cache
        idx[0] += 1
    return x + h


def attention_block(sd, p, x):
    """wan_video_vae.py:321-342 — per-frame single-head attention, head_dim = C."""
    b, c, t, h, w = x.shape
    y = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
    y = rms_norm_c(sd, p + ".norm", y)
    qkv = F.conv2d(y, sd[p + ".to_qkv.weight"], sd[p + ".to_qkv.bias"])
    q, k, v = qkv.reshape(b * t, 1, c * 3, -1).permute(0, 1, 3, 2).contiguous().chunk(3, dim=-1)
    y = F.scaled_dot_product_attention(q, k, v)
    y = y.squeeze(1).permute(0, 2, 1).reshape(b * t, c, h, w)
    y = F.conv2d(y, sd[p + ".proj.weight"], sd[p + ".proj.bias"])
    y = y.view(b, t, c, h, w).permute(0, 2, 1, 3, 4)
    return y + x


def resample_up(sd, p, x, cache, idx, temporal):
    """wan_video_vae.py:120-160 (+ Resample38 :242-252, Upsample :73-79)."""
    b, c, t, h, w = x.shape
    if temporal:
        i = idx[0]
        if cache[i] is None:
            cache[i] = "Rep"                       # first chunk: no temporal doubling
            idx[0] += 1
        else:
            new_cache = x[:, :, -CACHE_T:].clone()
            if new_cache.shape[2] < 2:
                if isinstance(cache[i], str):
                    new_cache = torch.cat([torch.zeros_like(new_cache), new_cache], dim=2)
                else:
                    new_cache = torch.cat([cache[i][:, :, -1:], new_cache], dim=2)
            prev = None if isinstance(cache[i], str) else cache[i]
            x = causal_conv3d(sd, p + ".time_conv", x, prev)
            cache[i] = new_cache
            idx[0] += 1
            x = x.reshape(b, 2, c, t, h, w)
            x = torch.stack((x[:, 0], x[:, 1]), 3).reshape(b, c, t * 2, h, w)
    t = x.shape[2]
    y = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
    y = F.interpolate(y.float(), scale_factor=(2.0, 2.0), mode="nearest-exact").type_as(y)
    y = F.conv2d(y, sd[p + ".resample.1.weight"], sd[p + ".resample.1.bias"], padding=1)
    return y.view(b, t, y.shape[1], 2 * h, 2 * w).permute(0, 2, 1, 3, 4)


def dup_up3d(x, out_channels, factor_t, factor_s, first_chunk):
    """wan_video_vae.py:417-439 — parameter-free repeat-interleave + pixel-shuffle shortcut."""
    b, c, t, h, w = x.shape
    factor = factor_t * factor_s * factor_s
    x = x.repeat_interleave(out_channels * factor // c, dim=1)
    x = x.view(b, out_channels, factor_t, factor_s, factor_s, t, h, w)
    x = x.permute(0, 1, 5, 2, 6, 3, 7, 4).contiguous()
    x = x.view(b, out_channels, t * factor_t, h * factor_s, w * factor_s)
    return x[:, :, factor_t - 1:] if first_chunk else x


def decoder_config(sd, prefix="model.decoder"):
    """Recover (dims, temporal flags) from tensor shapes, so tiny test decoders work too."""
    d0 = sd[prefix + ".conv1.weight"].shape[0]
    dims = [d0]
    i = 0
    while f"{prefix}.upsamples.{i}.upsamples.0.residual.2.weight" in sd:
        dims.append(sd[f"{prefix}.upsamples.{i}.upsamples.0.residual.2.weight"].shape[0])
        i += 1
    n_up = i
    temporal = [f"{prefix}.upsamples.{j}.upsamples.3.time_conv.weight" in sd for j in range(n_up)]
    has_up = [f"{prefix}.upsamples.{j}.upsamples.3.resample.1.weight" in sd for j in range(n_up)]
    return dims, temporal, has_up


def decoder_chunk(sd, x, cache, first_chunk, prefix="model.decoder"):
    """Decoder3d_38.forward, wan_video_vae.py:889-940, on one latent frame."""
    idx = [0]
    dims, temporal, has_up = decoder_config(sd, prefix)
    new_cache = _roll_cache(x, cache[0])
    x = causal_conv3d(sd, prefix + ".conv1", x, cache[0])
    cache[0] = new_cache
    idx[0] = 1
    x = residual_block(sd, prefix + ".middle.0", x, cache, idx)
    x = attention_block(sd, prefix + ".middle.1", x)
    x = residual_block(sd, prefix + ".middle.2", x, cache, idx)
    for j in range(len(dims) - 1):
        p = f"{prefix}.upsamples.{j}"
        main = x.clone()
        for r in range(3):
            main = residual_block(sd, f"{p}.upsamples.{r}", main, cache, idx)
        if has_up[j]:
            main = resample_up(sd, f"{p}.upsamples.3", main, cache, idx, temporal[j])
            x = main + dup_up3d(x, dims[j + 1], 2 if temporal[j] else 1, 2, first_chunk)
        else:
            x = main
    x = F.silu(rms_norm_c(sd, prefix + ".head.0", x))
    new_cache = _roll_cache(x, cache[idx[0]])
    x = causal_conv3d(sd, prefix + ".head.2", x, cache[idx[0]])
    cache[idx[0]] = new_cache
    idx[0] += 1
    return x


def count_causal_convs(sd, prefix="model.decoder"):
    """count_conv3d, wan_video_vae.py:943-948 — every CausalConv3d in the decoder (5-D weights)."""
    return sum(1 for k, v in sd.items() if k.startswith(prefix + ".") and k.endswith(".weight") and v.dim() == 5)


def unpatchify2(x):
    """wan_video_vae.py:214-224 with patch 2: 'b (c r q) f h w -> b c f (h q) (w r)'."""
    b, c4, f, h, w = x.shape
    c = c4 // 4
    x = x.view(b, c, 2, 2, f, h, w).permute(0, 1, 4, 5, 3, 6, 2)   # b c f h q w r
    return x.reshape(b, c, f, h * 2, w * 2)


def decode(sd, z, mean=None, std=None, prefix="model"):
    """VideoVAE38_.decode, wan_video_vae.py:1326-1351: z (1,48,T,h,w) -> (1,3,4T-3,16h,16w)."""
    zc = z.shape[1]
    mean = torch.tensor(VAE38_MEAN if mean is None else mean).to(z.dtype)
    inv_std = (1.0 / torch.tensor(VAE38_STD if std is None else std)).to(z.dtype)
    z = z / inv_std.view(1, zc, 1, 1, 1) + mean.view(1, zc, 1, 1, 1)
    x = causal_conv3d(sd, prefix + ".conv2", z)
    cache = [None] * count_causal_convs(sd, prefix + ".decoder")
    outs = []
    for i in range(z.shape[2]):
        outs.append(decoder_chunk(sd, x[:, :, i:i + 1], cache, first_chunk=(i == 0), prefix=prefix + ".decoder"))
    return unpatchify2(torch.cat(outs, dim=2))


# --------------------------------------------------------------------------------- tiled decode
def tile_tasks(H, W, tile_size, tile_stride):
    """wan_video_vae.py:1108-1115."""
    (sh, sw), (th, tw) = tile_size, tile_stride
    tasks = []
    for h in range(0, H, th):
        if h - th >= 0 and h - th + sh >= H:
            continue
        for w in range(0, W, tw):
            if w - tw >= 0 and w - tw + sw >= W:
                continue
            tasks.append((h, h + sh, w, w + sw))
    return tasks


def ramp_mask_1d(length, left_bound, right_bound, border):
    """wan_video_vae.py:1081-1087."""
    x = torch.ones((length,))
    if not left_bound:
        x[:border] = (torch.arange(border) + 1) / border
    if not right_bound:
        x[-border:] = torch.flip((torch.arange(border) + 1) / border, dims=(0,))
    return x


def tile_mask(Ht, Wt, is_bound, border):
    """wan_video_vae.py:1090-1100 — min of the two 1-D ramps, shape (1,1,1,Ht,Wt)."""
    mh = ramp_mask_1d(Ht, is_bound[0], is_bound[1], border[0])
    mw = ramp_mask_1d(Wt, is_bound[2], is_bound[3], border[1])
    return torch.minimum(mh[:, None].expand(Ht, Wt), mw[None, :].expand(Ht, Wt))[None, None, None]


def tiled_decode(sd, z, tile_size, tile_stride, upsampling=16, prefix="model"):
    """WanVideoVAE.tiled_decode, wan_video_vae.py:1103-1152 (accumulators in z.dtype, as there)."""
    _, _, T, H, W = z.shape
    out_T = T * 4 - 3
    weight = torch.zeros((1, 1, out_T, H * upsampling, W * upsampling), dtype=z.dtype)
    values = torch.zeros((1, 3, out_T, H * upsampling, W * upsampling), dtype=z.dtype)
    for h, h_, w, w_ in tile_tasks(H, W, tile_size, tile_stride):
        tile = decode(sd, z[:, :, :, h:h_, w:w_], prefix=prefix)
        m = tile_mask(tile.shape[3], tile.shape[4], (h == 0, h_ >= H, w == 0, w_ >= W),
                      ((tile_size[0] - tile_stride[0]) * upsampling, (tile_size[1] - tile_stride[1]) * upsampling))
        m = m.to(dtype=z.dtype)
        th, tw = h * upsampling, w * upsampling
        values[:, :, :, th:th + tile.shape[3], tw:tw + tile.shape[4]] += tile * m
        weight[:, :, :, th:th + tile.shape[3], tw:tw + tile.shape[4]] += m
    return (values / weight).clamp_(-1, 1)


def vae_decode(sd, latents, tiled=False, tile_size=(34, 34), tile_stride=(18, 16)):
    """WanVideoVAE.decode, wan_video_vae.py:1235-1247 — batch of latents -> (B,3,F,H,W) in [-1,1]."""
    vids = []
    for z in latents:
        z = z.unsqueeze(0)
        v = tiled_decode(sd, z, tile_size, tile_stride) if tiled else decode(sd, z).clamp_(-1, 1)
        vids.append(v.squeeze(0))
    return torch.stack(vids)


# ------------------------------------------------------------------------------------------- encoder (first-frame path)
def patchify2(x):
    """wan_video_vae.py:199-211 with patch 2: 'b c f (h q) (w r) -> b (c r q) f h w'."""
    b, c, f, H, W = x.shape
    x = x.view(b, c, f, H // 2, 2, W // 2, 2).permute(0, 1, 6, 4, 2, 3, 5)     # b c r q f h w
    return x.reshape(b, c * 4, f, H // 2, W // 2)


def avg_down3d(x, out_channels, factor_t, factor_s):
    """AvgDown3D.forward, wan_video_vae.py:363-395."""
    pad_t = (factor_t - x.shape[2] % factor_t) % factor_t
    x = F.pad(x, (0, 0, 0, 0, pad_t, 0))
    B, C, T, H, W = x.shape
    factor = factor_t * factor_s * factor_s
    x = x.view(B, C, T // factor_t, factor_t, H // factor_s, factor_s, W // factor_s, factor_s)
    x = x.permute(0, 1, 3, 5, 7, 2, 4, 6).contiguous()
    x = x.view(B, C * factor, T // factor_t, H // factor_s, W // factor_s)
    x = x.view(B, out_channels, C * factor // out_channels, T // factor_t, H // factor_s, W // factor_s)
    return x.mean(dim=2)


def resample_down(sd, p, x, cache, idx, temporal):
    """Resample.forward downsample2d / downsample3d, wan_video_vae.py:157-174 (+ Resample38 :253-263)."""
    b, c, t, h, w = x.shape
    y = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
    y = F.conv2d(F.pad(y, (0, 1, 0, 1)), sd[p + ".resample.1.weight"], sd[p + ".resample.1.bias"], stride=2)
    x = y.view(b, t, c, y.shape[2], y.shape[3]).permute(0, 2, 1, 3, 4)
    if temporal:
        i = idx[0]
        if cache[i] is None:
            cache[i] = x.clone()
            idx[0] += 1
        else:
            new_cache = x[:, :, -1:].clone()
            w_, b_ = sd[p + ".time_conv.weight"], sd[p + ".time_conv.bias"]
            x = F.conv3d(torch.cat([cache[i][:, :, -1:], x], 2), w_, b_, stride=(2, 1, 1))
            cache[i] = new_cache
            idx[0] += 1
    return x


def encoder_config(sd, prefix="model.encoder"):
    dims = [sd[prefix + ".conv1.weight"].shape[0]]
    i = 0
    while f"{prefix}.downsamples.{i}.downsamples.0.residual.2.weight" in sd:
        dims.append(sd[f"{prefix}.downsamples.{i}.downsamples.0.residual.2.weight"].shape[0])
        i += 1
    n = i
    has_down = [f"{prefix}.downsamples.{j}.downsamples.2.resample.1.weight" in sd for j in range(n)]
    temporal = [f"{prefix}.downsamples.{j}.downsamples.2.time_conv.weight" in sd for j in range(n)]
    return dims, has_down, temporal


def encoder_chunk(sd, x, cache, prefix="model.encoder"):
    """Encoder3d_38.forward, wan_video_vae.py:679-733."""
    idx = [0]
    dims, has_down, temporal = encoder_config(sd, prefix)
    new_cache = _roll_cache(x, cache[0])
    x = causal_conv3d(sd, prefix + ".conv1", x, cache[0])
    cache[0] = new_cache
    idx[0] = 1
    for j in range(len(dims) - 1):
        p = f"{prefix}.downsamples.{j}"
        x_copy = x.clone()
        for r in range(2):
            x = residual_block(sd, f"{p}.downsamples.{r}", x, cache, idx)
        if has_down[j]:
            x = resample_down(sd, f"{p}.downsamples.2", x, cache, idx, temporal[j])
        x = x + avg_down3d(x_copy, dims[j + 1], 2 if temporal[j] else 1, 2 if has_down[j] else 1)
    x = residual_block(sd, prefix + ".middle.0", x, cache, idx)
    x = attention_block(sd, prefix + ".middle.1", x)
    x = residual_block(sd, prefix + ".middle.2", x, cache, idx)
    x = F.silu(rms_norm_c(sd, prefix + ".head.0", x))
    new_cache = _roll_cache(x, cache[idx[0]])
    x = causal_conv3d(sd, prefix + ".head.2", x, cache[idx[0]])
    cache[idx[0]] = new_cache
    return x


def encode(sd, x, mean=None, std=None, prefix="model"):
    """VideoVAE38_.encode, wan_video_vae.py:1298-1323: video (1,3,T,H,W) in [-1,1] -> mu (1,48,(T+3)//4,H/16,W/16)."""
    x = patchify2(x)
    t = x.shape[2]
    cache = [None] * count_causal_convs(sd, prefix + ".encoder")
    outs = []
    for i in range(1 + (t - 1) // 4):
        chunk = x[:, :, :1] if i == 0 else x[:, :, 1 + 4 * (i - 1):1 + 4 * i]
        outs.append(encoder_chunk(sd, chunk, cache, prefix + ".encoder"))
    out = torch.cat(outs, 2)
    mu, _ = causal_conv3d(sd, prefix + ".conv1", out).chunk(2, dim=1)
    zc = mu.shape[1]
    mean = torch.tensor(VAE38_MEAN if mean is None else mean).to(mu.dtype)
    inv_std = (1.0 / torch.tensor(VAE38_STD if std is None else std)).to(mu.dtype)
    return (mu - mean.view(1, zc, 1, 1, 1)) * inv_std.view(1, zc, 1, 1, 1)


def tiled_encode(sd, video, tile_size, tile_stride, upsampling=16, prefix="model"):
    """WanVideoVAE.tiled_encode, wan_video_vae.py:1155-1203 (tile sizes in pixels)."""
    _, _, T, H, W = video.shape
    out_T = (T + 3) // 4
    zc = sd[prefix + ".conv2.weight"].shape[0]
    weight = torch.zeros((1, 1, out_T, H // upsampling, W // upsampling), dtype=video.dtype)
    values = torch.zeros((1, zc, out_T, H // upsampling, W // upsampling), dtype=video.dtype)
    for h, h_, w, w_ in tile_tasks(H, W, tile_size, tile_stride):
        tile = encode(sd, video[:, :, :, h:h_, w:w_], prefix=prefix)
        m = tile_mask(tile.shape[3], tile.shape[4], (h == 0, h_ >= H, w == 0, w_ >= W),
                      ((tile_size[0] - tile_stride[0]) // upsampling, (tile_size[1] - tile_stride[1]) // upsampling))
        m = m.to(dtype=video.dtype)
        th, tw = h // upsampling, w // upsampling
        values[:, :, :, th:th + tile.shape[3], tw:tw + tile.shape[4]] += tile * m
        weight[:, :, :, th:th + tile.shape[3], tw:tw + tile.shape[4]] += m
    return values / weight


def vae_encode(sd, videos, tiled=False, tile_size=(34, 34), tile_stride=(18, 16), upsampling=16):
    """WanVideoVAE.encode, wan_video_vae.py:1218-1232: list of (3,T,H,W) -> (B,48,T',h,w)."""
    outs = []
    for video in videos:
        video = video.unsqueeze(0)
        if tiled:
            ts = (tile_size[0] * upsampling, tile_size[1] * upsampling)
            st = (tile_stride[0] * upsampling, tile_stride[1] * upsampling)
            z = tiled_encode(sd, video, ts, st, upsampling)
        else:
            z = encode(sd, video)
        outs.append(z.squeeze(0))
    return torch.stack(outs)

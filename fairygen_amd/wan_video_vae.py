"""Wan2.2 VAE ("VAE38") on MI355X: reference parameter layout, decode on the HIP kernels.

Mirror of ``diffsynth/models/wan_video_vae.py``: ``WanVideoVAE38`` :1354-1382, ``VideoVAE38_`` :1269-1351,
``Decoder3d_38`` :842-940, ``Up_ResidualBlock`` :477-514, ``ResidualBlock`` :267-301, ``Resample38`` :227-265,
``AttentionBlock`` :304-342, ``DupUp3D`` :398-439, tiling ``WanVideoVAE`` :1081-1152,1235-1247.
The nn.Modules below are parameter containers with the reference's names/shapes (the encoder's included, so
``Wan2.2_VAE.pth`` loads and its key hash matches); the decode arithmetic is channels-last (T,H,W,C) on
``fairygen_amd.hip``: implicit-GEMM MFMA convs with the feature cache, fused RMS_norm·SiLU, fused nearest-2x
upsample, fused channel->time interleave, fused residual add, on-device tile feathering.
The chunk loop (one latent frame per step, 2-frame feature cache per causal conv) stays on the host, as in
the reference; every cache is a zero-initialised (2,H,W,C) ring that frames are shifted into, which is what
the reference's None / 1-frame / 'Rep' special cases amount to (zero causal padding).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import hip

CACHE_T = 2


class CausalConv3d(nn.Conv3d):
    """Parameter holder (weight (Cout,Cin,kt,kh,kw), bias); compute = hip.conv3d_cl."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._padding = (self.padding[2], self.padding[2], self.padding[1], self.padding[1], 2 * self.padding[0], 0)
        self.padding = (0, 0, 0)


class RMS_norm(nn.Module):
    def __init__(self, dim, channel_first=True, images=True, bias=False):
        super().__init__()
        shape = (dim, *((1, 1, 1) if not images else (1, 1))) if channel_first else (dim,)
        self.channel_first, self.scale = channel_first, dim ** 0.5
        self.gamma = nn.Parameter(torch.ones(shape))
        self.bias = nn.Parameter(torch.zeros(shape)) if bias else 0.0


class Upsample(nn.Upsample):
    pass


class Resample38(nn.Module):
    def __init__(self, dim, mode):
        assert mode in ("none", "upsample2d", "upsample3d", "downsample2d", "downsample3d")
        super().__init__()
        self.dim, self.mode = dim, mode
        if mode in ("upsample2d", "upsample3d"):
            self.resample = nn.Sequential(Upsample(scale_factor=(2.0, 2.0), mode="nearest-exact"),
                                          nn.Conv2d(dim, dim, 3, padding=1))
            if mode == "upsample3d":
                self.time_conv = CausalConv3d(dim, dim * 2, (3, 1, 1), padding=(1, 0, 0))
        elif mode in ("downsample2d", "downsample3d"):
            self.resample = nn.Sequential(nn.ZeroPad2d((0, 1, 0, 1)), nn.Conv2d(dim, dim, 3, stride=(2, 2)))
            if mode == "downsample3d":
                self.time_conv = CausalConv3d(dim, dim, (3, 1, 1), stride=(2, 1, 1), padding=(0, 0, 0))
        else:
            self.resample = nn.Identity()


class ResidualBlock(nn.Module):
    def __init__(self, in_dim, out_dim, dropout=0.0):
        super().__init__()
        self.in_dim, self.out_dim = in_dim, out_dim
        self.residual = nn.Sequential(
            RMS_norm(in_dim, images=False), nn.SiLU(), CausalConv3d(in_dim, out_dim, 3, padding=1),
            RMS_norm(out_dim, images=False), nn.SiLU(), nn.Dropout(dropout), CausalConv3d(out_dim, out_dim, 3, padding=1))
        self.shortcut = CausalConv3d(in_dim, out_dim, 1) if in_dim != out_dim else nn.Identity()


class AttentionBlock(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.dim = dim
        self.norm = RMS_norm(dim)
        self.to_qkv = nn.Conv2d(dim, dim * 3, 1)
        self.proj = nn.Conv2d(dim, dim, 1)


class AvgDown3D(nn.Module):
    def __init__(self, in_channels, out_channels, factor_t, factor_s=1):
        super().__init__()
        self.in_channels, self.out_channels, self.factor_t, self.factor_s = in_channels, out_channels, factor_t, factor_s


class DupUp3D(nn.Module):
    def __init__(self, in_channels, out_channels, factor_t, factor_s=1):
        super().__init__()
        self.in_channels, self.out_channels, self.factor_t, self.factor_s = in_channels, out_channels, factor_t, factor_s


class Down_ResidualBlock(nn.Module):
    def __init__(self, in_dim, out_dim, dropout, mult, temperal_downsample=False, down_flag=False):
        super().__init__()
        self.avg_shortcut = AvgDown3D(in_dim, out_dim, 2 if temperal_downsample else 1, 2 if down_flag else 1)
        layers = []
        for _ in range(mult):
            layers.append(ResidualBlock(in_dim, out_dim, dropout))
            in_dim = out_dim
        if down_flag:
            layers.append(Resample38(out_dim, mode="downsample3d" if temperal_downsample else "downsample2d"))
        self.downsamples = nn.Sequential(*layers)


class Up_ResidualBlock(nn.Module):
    def __init__(self, in_dim, out_dim, dropout, mult, temperal_upsample=False, up_flag=False):
        super().__init__()
        self.avg_shortcut = DupUp3D(in_dim, out_dim, 2 if temperal_upsample else 1, 2 if up_flag else 1) if up_flag else None
        layers = []
        for _ in range(mult):
            layers.append(ResidualBlock(in_dim, out_dim, dropout))
            in_dim = out_dim
        if up_flag:
            layers.append(Resample38(out_dim, mode="upsample3d" if temperal_upsample else "upsample2d"))
        self.upsamples = nn.Sequential(*layers)


class Encoder3d_38(nn.Module):
    """Parameters only (first-frame encode is the next §8(f) row; see WanVideoVAE38.encode)."""

    def __init__(self, dim=128, z_dim=4, dim_mult=[1, 2, 4, 4], num_res_blocks=2, attn_scales=[],
                 temperal_downsample=[False, True, True], dropout=0.0):
        super().__init__()
        dims = [dim * u for u in [1] + dim_mult]
        self.conv1 = CausalConv3d(12, dims[0], 3, padding=1)
        downs = []
        for i, (in_dim, out_dim) in enumerate(zip(dims[:-1], dims[1:])):
            t_down = temperal_downsample[i] if i < len(temperal_downsample) else False
            downs.append(Down_ResidualBlock(in_dim, out_dim, dropout, num_res_blocks, t_down, i != len(dim_mult) - 1))
        self.downsamples = nn.Sequential(*downs)
        self.middle = nn.Sequential(ResidualBlock(out_dim, out_dim, dropout), AttentionBlock(out_dim),
                                    ResidualBlock(out_dim, out_dim, dropout))
        self.head = nn.Sequential(RMS_norm(out_dim, images=False), nn.SiLU(), CausalConv3d(out_dim, z_dim, 3, padding=1))


class Decoder3d_38(nn.Module):
    def __init__(self, dim=128, z_dim=4, dim_mult=[1, 2, 4, 4], num_res_blocks=2, attn_scales=[],
                 temperal_upsample=[False, True, True], dropout=0.0):
        super().__init__()
        dims = [dim * u for u in [dim_mult[-1]] + dim_mult[::-1]]
        self.conv1 = CausalConv3d(z_dim, dims[0], 3, padding=1)
        self.middle = nn.Sequential(ResidualBlock(dims[0], dims[0], dropout), AttentionBlock(dims[0]),
                                    ResidualBlock(dims[0], dims[0], dropout))
        ups = []
        for i, (in_dim, out_dim) in enumerate(zip(dims[:-1], dims[1:])):
            t_up = temperal_upsample[i] if i < len(temperal_upsample) else False
            ups.append(Up_ResidualBlock(in_dim, out_dim, dropout, num_res_blocks + 1, t_up, i != len(dim_mult) - 1))
        self.upsamples = nn.Sequential(*ups)
        self.head = nn.Sequential(RMS_norm(out_dim, images=False), nn.SiLU(), CausalConv3d(out_dim, 12, 3, padding=1))


class _ConvState:
    """Packed weight of one conv + (for causal 3x3x3 / 3x1x1 convs) its input ring for one decode:
    ring[0:2] = the feature cache (the previous two input frames, zeros at the start), ring[2:2+T] = the chunk's
    input, written in place by the producer kernel, so the conv reads one contiguous (T+2,H,W,C) tensor."""

    __slots__ = ("packed", "bias", "cout", "cin", "kt", "ks", "ring")

    def __init__(self, conv):
        w = conv.weight
        self.packed = hip.conv_pack_weight(w)
        self.bias = conv.bias.contiguous()
        self.cout, self.cin = w.shape[0], w.shape[1]
        self.kt = w.shape[2] if w.dim() == 5 else 1
        self.ks = w.shape[-1]
        self.ring = None

    def slot(self, t, h, w, dtype, device):
        """View (t,h,w,cin) of the ring where the producer writes this chunk's input."""
        if self.ring is None or self.ring.shape[0] < t + CACHE_T or tuple(self.ring.shape[1:3]) != (h, w):
            old = self.ring
            cin = self.cin if self.cin % 8 == 0 else (self.cin + 15) // 16 * 16     # encoder conv1: 12 -> 16 (zero padded)
            self.ring = torch.zeros((t + CACHE_T, h, w, cin), dtype=dtype, device=device)
            if old is not None and tuple(old.shape[1:3]) == (h, w):
                self.ring[:CACHE_T].copy_(old[:CACHE_T])
        return self.ring[CACHE_T:CACHE_T + t]

    def shift(self, t):
        """After the conv: the last two frames of [cache; chunk] become the cache (wan_video_vae.py:288-297)."""
        self.ring[0].copy_(self.ring[t])
        self.ring[1].copy_(self.ring[t + 1])


class VideoVAE38_(nn.Module):
    def __init__(self, dim=160, z_dim=48, dec_dim=256, dim_mult=[1, 2, 4, 4], num_res_blocks=2, attn_scales=[],
                 temperal_downsample=[False, True, True], dropout=0.0):
        super().__init__()
        self.dim, self.dim_mult = dim, dim_mult
        self.temperal_downsample = temperal_downsample
        self.temperal_upsample = temperal_downsample[::-1]
        self.encoder = Encoder3d_38(dim, z_dim * 2, dim_mult, num_res_blocks, attn_scales, temperal_downsample, dropout)
        self.conv1 = CausalConv3d(z_dim * 2, z_dim * 2, 1)
        self.conv2 = CausalConv3d(z_dim, z_dim, 1)
        self.decoder = Decoder3d_38(dec_dim, z_dim, dim_mult, num_res_blocks, attn_scales, self.temperal_upsample, dropout)
        self._conv_states = None
        self._down_cache = {}
        self.z_dim = z_dim
        # latent frames per decoder call after the first (1 = the reference's chunking; identical arithmetic per output for any value).  8 is what
        # the conv kernel's 32-bit offsets allow at the (30, 52) tile; measured on it: 4 -> 808 ms, 6 -> 790, 8 -> 786 (fewer, larger launches)
        self.max_chunk_group = 8

    # ------------------------------------------------------------------ weight preparation
    def invalidate_packed(self):
        self._conv_states = None

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self.invalidate_packed()
        return out

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self.invalidate_packed()
        return out

    def _states(self, encoder=False):
        if self._conv_states is None:
            self._conv_states = {}
        probe = self.conv1 if encoder else self.conv2
        if id(probe) not in self._conv_states:      # decoder and encoder weights are packed on first use of each
            root = self.encoder if encoder else self.decoder
            for mod in [probe] + [m for m in root.modules() if isinstance(m, (CausalConv3d, nn.Conv2d))]:
                if mod.weight.shape[-1] == 1 and mod.weight.dim() == 4:
                    continue        # 1x1 Conv2d of the AttentionBlock: plain GEMMs (F.linear)
                self._conv_states[id(mod)] = _ConvState(mod)
        return self._conv_states

    # ------------------------------------------------------------------ building blocks (channels-last)
    def _conv(self, conv, x, residual=None, upsample2x=False, time_interleave=False, downsample2x=False):
        """Conv without temporal extent (1x1x1 shortcut / conv1 / conv2, the 3x3 Conv2d of the resamplers)."""
        st = self._states()[id(conv)]
        return hip.conv3d_cl(x, st.packed, st.bias, st.cout, st.kt, st.ks, residual=residual, upsample2x=upsample2x,
                             time_interleave=time_interleave, downsample2x=downsample2x)

    def _slot(self, conv, like, t=None):
        """Where the producer of `conv`'s input must write: a (T,H,W,Cin) view inside the conv's ring."""
        st = self._states()[id(conv)]
        return st.slot(like.shape[0] if t is None else t, like.shape[1], like.shape[2], like.dtype, like.device)

    def _cconv(self, conv, t, residual=None, time_interleave=False):
        """Causal conv over [feature cache; chunk] (its input was written into the ring slot), then roll the cache."""
        st = self._states()[id(conv)]
        y = hip.conv3d_cl(st.ring[: t + CACHE_T], st.packed, st.bias, st.cout, st.kt, st.ks, residual=residual,
                          time_interleave=time_interleave)
        st.shift(t)
        return y

    def _res(self, blk, x):
        r = blk.residual
        t = x.shape[0]
        h = x if isinstance(blk.shortcut, nn.Identity) else self._conv(blk.shortcut, x)
        hip.vae_rmsnorm_silu(x, r[0].gamma.view(-1), True, out=self._slot(r[2], x))
        y = self._cconv(r[2], t)
        hip.vae_rmsnorm_silu(y, r[3].gamma.view(-1), True, out=self._slot(r[6], y))
        return self._cconv(r[6], t, residual=h)

    def _attn(self, blk, x):
        t, hh, ww, c = x.shape
        xn = hip.vae_rmsnorm_silu(x, blk.norm.gamma.view(-1), False)
        qkv = F.linear(xn.view(t, hh * ww, c), blk.to_qkv.weight.view(3 * c, c), blk.to_qkv.bias)
        outs = []
        for f in range(t):          # single head, head_dim = C, per frame (:327-337); GEMMs on hipBLASLt
            q, k, v = qkv[f, :, :c], qkv[f, :, c:2 * c], qkv[f, :, 2 * c:]
            scores = torch.matmul(q.float(), k.float().t()).contiguous()
            outs.append(torch.matmul(hip.softmax_rows(scores, c ** -0.5), v))
        o = F.linear(torch.stack(outs), blk.proj.weight.view(c, c), blk.proj.bias)
        return hip.gate_residual(o.view(t, hh, ww, c).contiguous(), x)      # proj(x) + identity

    def _resample_up(self, rs, x, first_chunk):
        if rs.mode == "upsample3d" and not first_chunk:      # first chunk: 'Rep', no temporal doubling (:125-127)
            self._slot(rs.time_conv, x).copy_(x)
            x = self._cconv(rs.time_conv, x.shape[0], time_interleave=True)
        return self._conv(rs.resample[1], x, upsample2x=True)

    def _decoder_chunk(self, x, first_chunk):
        dec = self.decoder
        self._slot(dec.conv1, x).copy_(x)
        x = self._cconv(dec.conv1, x.shape[0])
        x = self._res(dec.middle[0], x)
        x = self._attn(dec.middle[1], x)
        x = self._res(dec.middle[2], x)
        for up in dec.upsamples:
            layers = list(up.upsamples)
            main = x
            for blk in layers[:3]:
                main = self._res(blk, main)
            if up.avg_shortcut is not None:
                main = self._resample_up(layers[3], main, first_chunk)
                sc = up.avg_shortcut
                x = hip.dupup3d_add(x, main, sc.out_channels, sc.factor_t, sc.factor_s, first_chunk)
            else:
                x = main
        hip.vae_rmsnorm_silu(x, dec.head[0].gamma.view(-1), True, out=self._slot(dec.head[2], x))
        return self._cconv(dec.head[2], x.shape[0])

    def clear_cache(self):
        self._down_cache = {}
        if self._conv_states is not None:
            for st in self._conv_states.values():
                st.ring = None

    def decode(self, z, scale, clamp=False):
        """z (1,48,T,h,w) on the HIP device -> (1,3,4T-3,16h,16w) (reference :1326-1351)."""
        assert z.dim() == 5 and z.shape[0] == 1, "decode one latent at a time (as the reference's tile loop does)"
        self.clear_cache()
        mean, inv_std = (s.to(dtype=z.dtype, device=z.device) for s in scale)
        _, _, T, h, w = z.shape
        x = hip.vae_latent_to_cl(z[0].contiguous(), mean.contiguous(), inv_std.contiguous())
        x = self._conv(self.conv2, x)
        frames = 4 * T - 3
        video = torch.empty((3, frames, 16 * h, 16 * w), dtype=z.dtype, device=z.device)
        # The reference decodes one latent frame per decoder call (:1337-1348).  Every op after the first chunk is
        # causal in time with the 2-frame feature cache, so latent frames 1.. can go through the decoder in groups
        # with identical arithmetic per output; larger groups fill the chip on the low-resolution layers.  The
        # group is bounded by the 32-bit byte offsets of the conv kernel (largest ring: (4G+2, 8h, 8w, widest C)).
        widest = max(st.cin for st in self._states().values())
        group = int(max(1, min(self.max_chunk_group, (3.5e9 / (8 * h * 8 * w * min(widest, 512) * 2) - 2) // 4)))
        t0, i = 0, 0
        while i < T:
            n = 1 if i == 0 else min(group, T - i)
            out = self._decoder_chunk(x[i:i + n], first_chunk=(i == 0))
            hip.vae_unpatchify(out, video, t0, clamp)
            t0 += out.shape[0]
            i += n
        assert t0 == frames
        self.clear_cache()
        return video.unsqueeze(0)

    def _resample_down(self, rs, x, first_chunk):
        """Resample.forward, downsample modes (reference :155-174): strided 3x3 Conv2d per frame, then for 'downsample3d' the
        temporal stride-2 time_conv over [last frame of the previous chunk; chunk].  The first chunk only leaves its frame
        behind as that cache and skips time_conv (:163-165)."""
        x = self._conv(rs.resample[1], x, downsample2x=True)
        if rs.mode == "downsample3d":
            key = id(rs)
            if first_chunk:
                self._down_cache[key] = x[-1:].clone()
            else:
                seq = torch.cat([self._down_cache[key], x], 0)            # (t+1, h, w, c); kernel (3,1,1), no padding
                self._down_cache[key] = x[-1:].clone()
                x = self._conv(rs.time_conv, seq)[::2].contiguous()        # stride 2 in time = every other stride-1 output
        return x

    def _encoder_chunk(self, y, first_chunk):
        """One Encoder3d_38 pass (reference :673-733) over a chunk of patchified frames (1 frame first, then 4)."""
        enc = self.encoder
        self._slot(enc.conv1, y).copy_(y)
        y = self._cconv(enc.conv1, y.shape[0])
        for blk in enc.downsamples:
            y_copy = y
            for layer in blk.downsamples:
                y = self._res(layer, y) if isinstance(layer, ResidualBlock) else self._resample_down(layer, y, first_chunk)
            sc = blk.avg_shortcut
            y = hip.avgdown3d_add(y_copy, y, sc.factor_t, sc.factor_s)
        y = self._res(enc.middle[0], y)
        y = self._attn(enc.middle[1], y)
        y = self._res(enc.middle[2], y)
        hip.vae_rmsnorm_silu(y, enc.head[0].gamma.view(-1), True, out=self._slot(enc.head[2], y))
        return self._cconv(enc.head[2], y.shape[0])

    def encode(self, x, scale):
        """x (1,3,T,H,W) in [-1,1] on the HIP device -> mu (1,48,1+(T-1)//4,H/16,W/16), normalised (reference :1298-1323):
        the first frame alone, then chunks of 4 frames through the encoder with its causal feature caches.  T = 1 is the
        TI2V first-frame conditioning of the FairyGen path; T > 1 is the video-to-video input."""
        assert x.dim() == 5 and x.shape[0] == 1 and x.shape[1] == 3
        self._states(encoder=True)
        self.clear_cache()
        mean, inv_std = (s.to(dtype=x.dtype, device=x.device).contiguous() for s in scale)
        frames = hip.vae_patchify(x[0].contiguous())
        outs = []
        for i in range(1 + (x.shape[2] - 1) // 4):
            chunk = frames[:1] if i == 0 else frames[1 + 4 * (i - 1):1 + 4 * i]
            outs.append(self._encoder_chunk(chunk, first_chunk=(i == 0)))
        y = self._conv(self.conv1, outs[0] if len(outs) == 1 else torch.cat(outs, 0))
        mu = hip.vae_latent_from_cl(y, mean, inv_std, self.z_dim)
        self.clear_cache()
        return mu.unsqueeze(0)


class WanVideoVAE38(nn.Module):
    MEAN = [-0.2289, -0.0052, -0.1323, -0.2339, -0.2799, 0.0174, 0.1838, 0.1557, -0.1382, 0.0542, 0.2813, 0.0891,
            0.1570, -0.0098, 0.0375, -0.1825, -0.2246, -0.1207, -0.0698, 0.5109, 0.2665, -0.2108, -0.2158, 0.2502,
            -0.2055, -0.0322, 0.1109, 0.1567, -0.0729, 0.0899, -0.2799, -0.1230, -0.0313, -0.1649, 0.0117, 0.0723,
            -0.2839, -0.2083, -0.0520, 0.3748, 0.0152, 0.1957, 0.1433, -0.2944, 0.3573, -0.0548, -0.1681, -0.0667]
    STD = [0.4765, 1.0364, 0.4514, 1.1677, 0.5313, 0.4990, 0.4818, 0.5013, 0.8158, 1.0344, 0.5894, 1.0901,
           0.6885, 0.6165, 0.8454, 0.4978, 0.5759, 0.3523, 0.7135, 0.6804, 0.5833, 1.4146, 0.8986, 0.5659,
           0.7069, 0.5338, 0.4889, 0.4917, 0.4069, 0.4999, 0.6866, 0.4093, 0.5709, 0.6065, 0.6415, 0.4944,
           0.5726, 1.2042, 0.5458, 1.6887, 0.3971, 1.0600, 0.3943, 0.5537, 0.5444, 0.4089, 0.7468, 0.7744]

    def __init__(self, z_dim=48, dim=160, dec_dim=256):
        super().__init__()
        self.mean = torch.tensor(self.MEAN, device="cpu")     # explicit cpu: models are built under device("meta")
        self.std = torch.tensor(self.STD, device="cpu")
        self.scale = [self.mean, 1.0 / self.std]
        self.model = VideoVAE38_(z_dim=z_dim, dim=dim, dec_dim=dec_dim).eval().requires_grad_(False)
        self.upsampling_factor = 16
        self.z_dim = z_dim

    @staticmethod
    def tile_tasks(H, W, tile_size, tile_stride):
        """Tile grid of tiled_decode (:1108-1115)."""
        (size_h, size_w), (stride_h, stride_w) = tile_size, tile_stride
        tasks = []
        for h in range(0, H, stride_h):
            if h - stride_h >= 0 and h - stride_h + size_h >= H:
                continue
            for w in range(0, W, stride_w):
                if w - stride_w >= 0 and w - stride_w + size_w >= W:
                    continue
                tasks.append((h, h + size_h, w, w + size_w))
        return tasks

    def tiled_decode(self, hidden_states, device, tile_size, tile_stride, shard=None):
        """Same tiles, masks and bf16 accumulation ORDER as the reference (:1103-1152), but the canvas lives in HBM
        (654 MB at 704x1280x121) instead of bouncing every tile through host memory.  With a multi-rank `shard`
        (sequence_parallel.TokenShard) the tiles are dealt to ranks by area (sequence_parallel.assign_tiles), decoded
        and broadcast; every rank then blends all tiles in the reference's order, so the result is identical to the
        single-GPU one."""
        _, _, T, H, W = hidden_states.shape
        up = self.upsampling_factor
        out_T = T * 4 - 3
        z = hidden_states.to(device)
        weight = torch.zeros((1, 1, out_T, H * up, W * up), dtype=z.dtype, device=device)
        values = torch.zeros((1, 3, out_T, H * up, W * up), dtype=z.dtype, device=device)
        tasks = self.tile_tasks(H, W, tile_size, tile_stride)
        world, rank = (shard.world_size, shard.rank) if shard is not None else (1, 0)
        exchange = shard is not None and shard.active
        if exchange:
            from .sequence_parallel import assign_tiles
            owner = assign_tiles([(min(h_, H) - h) * (min(w_, W) - w) for h, h_, w, w_ in tasks], world)
        else:
            owner = [0] * len(tasks)
        tiles = {}
        for i, (h, h_, w, w_) in enumerate(tasks):
            if owner[i] == rank:
                tiles[i] = self.model.decode(z[:, :, :, h:h_, w:w_].contiguous(), self.scale)
        for i, (h, h_, w, w_) in enumerate(tasks):
            if exchange:
                th, tw = (min(h_, H) - h) * up, (min(w_, W) - w) * up
                tile = tiles.pop(i) if i in tiles else torch.empty((1, 3, out_T, th, tw), dtype=z.dtype, device=device)
                shard.broadcast(tile, src=owner[i])
            else:
                tile = tiles.pop(i)
            hip.vae_tile_accumulate(tile[0], values[0], weight[0, 0], h * up, w * up,
                                    (tile_size[0] - tile_stride[0]) * up, (tile_size[1] - tile_stride[1]) * up,
                                    (h == 0, h_ >= H, w == 0, w_ >= W))
            del tile
        hip.vae_tile_finalize(values[0], weight[0, 0])
        return values

    def single_decode(self, hidden_state, device):
        return self.model.decode(hidden_state.to(device), self.scale, clamp=True)

    def decode(self, hidden_states, device, tiled=False, tile_size=(34, 34), tile_stride=(18, 16), shard=None):
        """(B,48,T,h,w) -> (B,3,F,H,W) in [-1,1], kept on `device` (:1235-1247)."""
        videos = []
        for hidden_state in hidden_states:
            hidden_state = hidden_state.unsqueeze(0)
            video = self.tiled_decode(hidden_state, device, tile_size, tile_stride, shard) if tiled \
                else self.single_decode(hidden_state, device)
            videos.append(video.squeeze(0))
        return torch.stack(videos)

    def tiled_encode(self, video, device, tile_size, tile_stride):
        """(1,3,T,H,W) -> (1,48,T',H/16,W/16): pixel-space tiles, latent-space feathering (:1155-1203)."""
        _, _, T, H, W = video.shape
        up = self.upsampling_factor
        out_T = (T + 3) // 4
        video = video.to(device)
        weight = torch.zeros((1, 1, out_T, H // up, W // up), dtype=video.dtype, device=device)
        values = torch.zeros((1, self.z_dim, out_T, H // up, W // up), dtype=video.dtype, device=device)
        for h, h_, w, w_ in self.tile_tasks(H, W, tile_size, tile_stride):
            tile = self.model.encode(video[:, :, :, h:h_, w:w_].contiguous(), self.scale)
            hip.vae_tile_accumulate(tile[0], values[0], weight[0, 0], h // up, w // up,
                                    (tile_size[0] - tile_stride[0]) // up, (tile_size[1] - tile_stride[1]) // up,
                                    (h == 0, h_ >= H, w == 0, w_ >= W))
        hip.vae_tile_finalize(values[0], weight[0, 0], clamp=False)
        return values

    def encode(self, videos, device, tiled=False, tile_size=(34, 34), tile_stride=(18, 16)):
        """list of (3,T,H,W) in [-1,1] -> (B,48,T',H/16,W/16) on `device` (:1218-1232)."""
        outs = []
        for video in videos:
            video = video.unsqueeze(0).to(device)
            if tiled:
                up = self.upsampling_factor
                z = self.tiled_encode(video, device, (tile_size[0] * up, tile_size[1] * up), (tile_stride[0] * up, tile_stride[1] * up))
            else:
                z = self.model.encode(video, self.scale)
            outs.append(z.squeeze(0))
        return torch.stack(outs)

// HBM-bound VAE decode helpers (channels-last activations): RMS_norm(+SiLU), DupUp3D shortcut + add,
// row softmax for the mid AttentionBlock, layout boundary kernels, tiled-decode feathering, uint8 frames.
#include "common.h"
#include <type_traits>

namespace {

constexpr int kMaxVecV = 4;   // C <= 2048 per pixel row, one wave per pixel


// F.normalize(x, dim=C) * sqrt(C) * gamma (+ SiLU): every op rounded to bf16 like the reference chain.
// LPP lanes own one pixel (64: C <= 2048; 32: C <= 256, two pixels per wave so that no lane idles on the 256-channel
// last stage, where this kernel is VALU-bound: the divisions, roundings and SiLU of every element).
// a / b from r = rcp(b) (1 ulp) and one residual correction: q0 = a r, q = q0 + (a - b q0) r — within 1 fp32 ulp of the IEEE quotient (every result here
// is rounded to bf16 next, 16 bits coarser), for ~3 VALU operations instead of the ~10 of the full division sequence: this kernel is VALU-bound
// (measured 3.2 TB/s on the 256-channel stage with two IEEE divisions per element)
__device__ __forceinline__ float div_by(float a, float b, float r) {
    const float q0 = a * r;
    return __builtin_fmaf(__builtin_fmaf(-b, q0, a), r, q0);
}

template <int LPP>
__global__ __launch_bounds__(256) void vae_rmsnorm_kernel(const bf16* __restrict__ x, const bf16* __restrict__ gamma,
                                                          bf16* __restrict__ out, int64_t pixels, int C, float sqrt_c,
                                                          int apply_silu) {
    constexpr int kPerWave = 64 / LPP, kVec = LPP == 64 ? kMaxVecV : 1;
    const int lane = threadIdx.x & (LPP - 1);
    const int64_t row = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * kPerWave + ((threadIdx.x & 63) / LPP);
    const bool live = row < pixels;          // both halves of a wave take part in the shuffles
    const int nvec = C >> 3;
    float v[kVec][8];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < kVec; ++i) {
        const int vi = lane + i * LPP;
        if (live && vi < nvec) {
            const bf16x8 t = *reinterpret_cast<const bf16x8*>(x + row * C + (int64_t)vi * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) { v[i][j] = (float)t[j]; ss += v[i][j] * v[i][j]; }
        }
    }
#pragma unroll
    for (int o = LPP / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    if (!live) return;
    const float denom = fmaxf(rbf(sqrtf(ss)), 1e-12f);
    const float rdenom = __builtin_amdgcn_rcpf(denom);
#pragma unroll
    for (int i = 0; i < kVec; ++i) {
        const int vi = lane + i * LPP;
        if (vi < nvec) {
            const bf16x8 g = *reinterpret_cast<const bf16x8*>(gamma + (int64_t)vi * 8);
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float y = rbf(rbf(rbf(div_by(v[i][j], denom, rdenom)) * sqrt_c) * (float)g[j]);
                if (apply_silu) {      // y / (1 + exp(-y)), the quotient again by reciprocal + one correction step
                    const float d = 1.0f + __expf(fminf(-y, 80.0f));      // finite d: rcp(inf) = 0 would turn the correction step into inf * 0
                    y = div_by(y, d, __builtin_amdgcn_rcpf(d));
                }
                o[j] = (bf16)y;
            }
            *reinterpret_cast<bf16x8*>(out + row * C + (int64_t)vi * 8) = o;
        }
    }
}

// out[t',y',x',oc] = main + x[t, y, x, (oc*factor + a*fs*fs + b*fs + c) / repeats]
//   with t' = t*ft + a - drop, y' = y*fs + b, x' = x*fs + c   (DupUp3D, models/wan_video_vae.py:417-439)
__global__ __launch_bounds__(256) void dupup3d_add_kernel(const bf16* __restrict__ x, const bf16* __restrict__ main_path,
                                                          bf16* __restrict__ out, int T, int H, int W, int Cin, int Cout,
                                                          int ft, int fs, int drop, int rep_shift) {
    // grid: x over (output column, 8-channel vector), y = output row, z = output frame: no division reaches 64 bits, none is per element
    const int Ho = H * fs, Wo = W * fs;
    const int factor = ft * fs * fs;
    const int repeats = Cout * factor / Cin;
    const int cvec = Cout >> 3;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= Wo * cvec) return;
    const int xo = idx / cvec, cv = idx - xo * cvec;
    const int yo = blockIdx.y, to = (int)blockIdx.z + drop;
    const int t = to / ft, a = to - t * ft, y = yo / fs, b = yo - y * fs, xx = xo / fs, c = xo - xx * fs;
    const int sub = a * fs * fs + b * fs + c;
    const bf16* src = x + (((int64_t)t * H + y) * W + xx) * Cin;
    const int64_t i = (((int64_t)blockIdx.z * Ho + yo) * Wo + xo) * cvec + cv;
    const bf16x8 m = *reinterpret_cast<const bf16x8*>(main_path + i * 8);
    bf16x8 o;
    const int stride = factor / repeats;      // repeats | factor (every DupUp3D of the VAE38 decoder): source channel = oc * stride + sub / repeats
    if (stride * repeats == factor && stride <= 4 && (Cin & 7) == 0) {
        // the 8 source channels of this vector sit in `stride` consecutive 16-byte vectors: vector loads instead of 8 two-byte gathers
        bf16 sv[32];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (q < stride) *reinterpret_cast<bf16x8*>(sv + 8 * q) = *reinterpret_cast<const bf16x8*>(src + (cv * stride + q) * 8);
        const int off = sub / repeats;
        // element j takes sv[j * stride + off]: written with compile-time indices (a run-time index would push sv into scratch memory)
        auto pick = [&](auto stride_c) {
            constexpr int S = decltype(stride_c)::value;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float v = (float)sv[j * S];
#pragma unroll
                for (int k = 1; k < S; ++k) v = off == k ? (float)sv[j * S + k] : v;
                o[j] = (bf16)((float)m[j] + v);
            }
        };
        if (stride == 1) pick(std::integral_constant<int, 1>{});
        else if (stride == 2) pick(std::integral_constant<int, 2>{});
        else if (stride == 3) pick(std::integral_constant<int, 3>{});
        else pick(std::integral_constant<int, 4>{});
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int num = (cv * 8 + j) * factor + sub;
            const float s1 = (float)src[rep_shift >= 0 ? (num >> rep_shift) : (num / repeats)];
            o[j] = (bf16)((float)m[j] + s1);
        }
    }
    *reinterpret_cast<bf16x8*>(out + i * 8) = o;
}

// one 256-thread block per row: probs = softmax(scores*scale) rounded to bf16
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ scores, bf16* __restrict__ probs,
                                                           int64_t cols, float scale) {
    __shared__ float red[8];
    const int64_t row = blockIdx.x;
    const float* s = scores + row * cols;
    float mx = -INFINITY;
    for (int64_t c = threadIdx.x; c < cols; c += 256) mx = fmaxf(mx, s[c]);
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int64_t c = threadIdx.x; c < cols; c += 256) sum += __expf((s[c] - mx) * scale);
    sum = wave_sum(sum);
    if ((threadIdx.x & 63) == 0) red[4 + (threadIdx.x >> 6)] = sum;
    __syncthreads();
    sum = red[4] + red[5] + red[6] + red[7];
    const float inv = 1.0f / sum;
    for (int64_t c = threadIdx.x; c < cols; c += 256) probs[row * cols + c] = (bf16)(__expf((s[c] - mx) * scale) * inv);
}

// (C,T,H,W) -> (T,H,W,C) with z/inv_std + mean, both steps rounded to bf16 (models/wan_video_vae.py:1330)
__global__ __launch_bounds__(256) void latent_to_cl_kernel(const bf16* __restrict__ z, const bf16* __restrict__ mean,
                                                           const bf16* __restrict__ inv_std, bf16* __restrict__ out, int C,
                                                           int64_t thw) {
    const int64_t total = thw * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t pix = i / C;
        const float v = (float)z[(int64_t)c * thw + pix];
        out[i] = (bf16)(rbf(v / (float)inv_std[c]) + (float)mean[c]);
    }
}

// (T,H,W,12) -> frames [t0,t0+T) of (3,F,2H,2W): channel = c*4 + r*2 + q -> out[c, t, 2y+q, 2x+r]
__global__ __launch_bounds__(256) void unpatchify_kernel(const bf16* __restrict__ x, bf16* __restrict__ video, int T, int H,
                                                         int W, int F, int t0, int do_clamp) {
    const int Ho = 2 * H, Wo = 2 * W;
    const int64_t total = (int64_t)3 * T * Ho * Wo;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int xo = (int)(i % Wo);
        int64_t rest = i / Wo;
        const int yo = (int)(rest % Ho); rest /= Ho;
        const int t = (int)(rest % T);
        const int c = (int)(rest / T);
        const int ch = c * 4 + (xo & 1) * 2 + (yo & 1);
        float v = (float)x[(((int64_t)t * H + (yo >> 1)) * W + (xo >> 1)) * 12 + ch];
        if (do_clamp) v = fminf(fmaxf(v, -1.f), 1.f);
        video[(((int64_t)c * F + t0 + t) * Ho + yo) * Wo + xo] = (bf16)v;
    }
}

__device__ __forceinline__ float ramp(int i, int len, int border, bool lo_bound, bool hi_bound) {
    float m = 1.f;
    if (!lo_bound && i < border) m = (float)(i + 1) / (float)border;
    if (!hi_bound && i >= len - border) m = (float)(len - i) / (float)border;   // flipped (arange+1)/border
    return m;
}

__global__ __launch_bounds__(256) void tile_accumulate_kernel(const bf16* __restrict__ tile, bf16* __restrict__ values,
                                                              bf16* __restrict__ weight, int C, int F, int Hv, int Wv, int th,
                                                              int tw, int y0, int x0, int bh, int bw, int bound) {
    const int64_t total = (int64_t)F * th * tw;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int xx = (int)(i % tw);
        const int yy = (int)((i / tw) % th);
        const int f = (int)(i / ((int64_t)tw * th));
        // both ramps are exact in fp32 as (k)/border; the reference builds them in fp32 then casts the min to bf16
        const float mh = ramp(yy, th, bh, bound & 1, bound & 2);
        const float mw = ramp(xx, tw, bw, bound & 4, bound & 8);
        const float m = rbf(fminf(mh, mw));
        const int64_t vo = ((int64_t)f * Hv + y0 + yy) * Wv + x0 + xx;
        for (int c = 0; c < C; ++c) {
            const int64_t vi = (int64_t)c * F * Hv * Wv + vo;
            const float t = (float)tile[((int64_t)c * F + f) * th * tw + (int64_t)yy * tw + xx];
            values[vi] = (bf16)((float)values[vi] + rbf(t * m));
        }
        weight[vo] = (bf16)((float)weight[vo] + m);
    }
}

__global__ __launch_bounds__(256) void tile_finalize_kernel(bf16* __restrict__ values, const bf16* __restrict__ weight,
                                                            int C, int64_t fhw, int do_clamp) {
    const int64_t total = C * fhw;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = rbf((float)values[i] / (float)weight[i % fhw]);
        values[i] = (bf16)(do_clamp ? fminf(fmaxf(v, -1.f), 1.f) : v);
    }
}

// (3,T,H,W) -> patchify(.,2) 'c f (h q) (w r) -> (c r q) f h w' (models/wan_video_vae.py:199-211), channels-last out
// padded to 16 channels (12 real + 4 zeros) so the first conv reads whole 16-byte chunks
__global__ __launch_bounds__(256) void patchify_kernel(const bf16* __restrict__ video, bf16* __restrict__ out, int T, int H,
                                                       int W) {      // H, W: patched (half) resolution
    const int64_t total = (int64_t)T * H * W * 16;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ch = (int)(i & 15);
        int64_t pix = i >> 4;
        const int x = (int)(pix % W); pix /= W;
        const int y = (int)(pix % H);
        const int t = (int)(pix / H);
        const int c = ch >> 2, r = (ch >> 1) & 1, q = ch & 1;
        out[i] = ch < 12 ? video[(((int64_t)c * T + t) * (2 * H) + 2 * y + q) * (2 * W) + 2 * x + r] : (bf16)0.f;
    }
}

// out = main + AvgDown3D(x): models/wan_video_vae.py:363-395,474.  x (T,H,W,Cin); main/out (To,H/fs,W/fs,Cout), To = ceil(T/ft);
// a missing leading frame (T not a multiple of ft) counts as zeros, exactly like the reference's front padding.
__global__ __launch_bounds__(256) void avgdown3d_add_kernel(const bf16* __restrict__ x, const bf16* __restrict__ main_path,
                                                            bf16* __restrict__ out, int T, int H, int W, int Cin, int Cout,
                                                            int ft, int fs) {
    const int To = (T + ft - 1) / ft, Ho = H / fs, Wo = W / fs, pad_t = To * ft - T;
    const int factor = ft * fs * fs, group = Cin * factor / Cout;
    const int64_t total = (int64_t)To * Ho * Wo * Cout;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int oc = (int)(i % Cout);
        int64_t pix = i / Cout;
        const int xo = (int)(pix % Wo); pix /= Wo;
        const int yo = (int)(pix % Ho);
        const int to = (int)(pix / Ho);
        float acc = 0.f;
        for (int gi = 0; gi < group; ++gi) {
            const int cc = oc * group + gi;                 // index into (c, a, b, d) = C x ft x fs x fs
            const int d = cc % fs, b = (cc / fs) % fs, a = (cc / (fs * fs)) % ft, c = cc / factor;
            const int t = to * ft + a - pad_t;
            if (t >= 0) acc += (float)x[(((int64_t)t * H + yo * fs + b) * W + xo * fs + d) * Cin + c];
        }
        out[i] = (bf16)((float)main_path[i] + rbf(acc / (float)group));
    }
}

// encoder tail: x (T,h,w,2z) channels-last -> mu = first z channels, (mu - mean) * inv_std, out (z,T,h,w)
__global__ __launch_bounds__(256) void latent_from_cl_kernel(const bf16* __restrict__ x, const bf16* __restrict__ mean,
                                                             const bf16* __restrict__ inv_std, bf16* __restrict__ out, int Z,
                                                             int Cx, int64_t thw) {
    const int64_t total = thw * Z;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = i % thw;
        const int c = (int)(i / thw);
        out[i] = (bf16)(rbf((float)x[pix * Cx + c] - (float)mean[c]) * (float)inv_std[c]);
    }
}

// (3,F,H,W) bf16 -> (F,H,W,3) uint8: ((x+1)*127.5).clip(0,255) in bf16 steps, truncation
__global__ __launch_bounds__(256) void video_to_uint8_kernel(const bf16* __restrict__ video, uint8_t* __restrict__ out,
                                                             int64_t fhw) {
    const int64_t total = fhw * 3;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % 3);
        const int64_t pix = i / 3;
        float v = rbf(rbf((float)video[(int64_t)c * fhw + pix] + 1.0f) * 127.5f);
        v = fminf(fmaxf(v, 0.f), 255.f);
        out[i] = (uint8_t)v;
    }
}

inline unsigned grid_for(int64_t n, int per_block = 256, int cap = 16384) {
    const int64_t g = (n + per_block - 1) / per_block;
    return (unsigned)(g < cap ? (g > 0 ? g : 1) : cap);
}

}  // namespace

extern "C" {

int fg_vae_rmsnorm_silu_bf16(const void* x, const void* gamma, void* out, int64_t pixels, int C, int apply_silu,
                             fg_stream_t stream) {
    FG_CHECK_ARG(x && gamma && out, "fg_vae_rmsnorm_silu_bf16: null pointer");
    FG_CHECK_ARG(pixels >= 0 && C > 0 && C % 8 == 0 && C <= kMaxVecV * 512, "fg_vae_rmsnorm_silu_bf16: need C %% 8 == 0, C <= %d",
                 kMaxVecV * 512);
    FG_CHECK_ARG(FG_ALIGNED16(x) && FG_ALIGNED16(gamma) && FG_ALIGNED16(out), "fg_vae_rmsnorm_silu_bf16: misaligned pointer");
    if (pixels == 0) return FG_OK;
    if (C <= 256)
        hipLaunchKernelGGL(vae_rmsnorm_kernel<32>, dim3((unsigned)((pixels + 7) / 8)), dim3(256), 0, (hipStream_t)stream,
                           (const bf16*)x, (const bf16*)gamma, (bf16*)out, pixels, C, sqrtf((float)C), apply_silu);
    else
        hipLaunchKernelGGL(vae_rmsnorm_kernel<64>, dim3((unsigned)((pixels + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                           (const bf16*)x, (const bf16*)gamma, (bf16*)out, pixels, C, sqrtf((float)C), apply_silu);
    return fg_launch_status("fg_vae_rmsnorm_silu_bf16");
}

int fg_dupup3d_add_bf16(const void* x, const void* main_path, void* out, int T, int H, int W, int Cin, int Cout, int ft,
                        int fs, int first_chunk, fg_stream_t stream) {
    FG_CHECK_ARG(x && main_path && out, "fg_dupup3d_add_bf16: null pointer");
    FG_CHECK_ARG(T > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Cout % 8 == 0 && (ft == 1 || ft == 2) && (fs == 1 || fs == 2),
                 "fg_dupup3d_add_bf16: bad sizes");
    FG_CHECK_ARG((Cout * ft * fs * fs) % Cin == 0, "fg_dupup3d_add_bf16: out_channels*factor must be divisible by in_channels");
    FG_CHECK_ARG(FG_ALIGNED16(main_path) && FG_ALIGNED16(out), "fg_dupup3d_add_bf16: misaligned pointer");
    const int drop = first_chunk ? ft - 1 : 0;
    const int To = T * ft - drop, Ho = H * fs, Wo = W * fs, repeats = Cout * ft * fs * fs / Cin;
    if (To <= 0) return FG_OK;
    FG_CHECK_ARG(Ho <= 65535 && To <= 65535 && (int64_t)Wo * (Cout / 8) < (1ll << 31), "fg_dupup3d_add_bf16: output too large for the grid");
    int rep_shift = -1;
    for (int sft = 0; sft < 16; ++sft)
        if ((1 << sft) == repeats) rep_shift = sft;
    hipLaunchKernelGGL(dupup3d_add_kernel, dim3((unsigned)((Wo * (Cout / 8) + 255) / 256), (unsigned)Ho, (unsigned)To), dim3(256), 0,
                       (hipStream_t)stream, (const bf16*)x, (const bf16*)main_path, (bf16*)out, T, H, W, Cin, Cout, ft, fs, drop, rep_shift);
    return fg_launch_status("fg_dupup3d_add_bf16");
}

int fg_softmax_rows_f32_bf16(const float* scores, void* probs, int64_t rows, int64_t cols, float scale, fg_stream_t stream) {
    FG_CHECK_ARG(scores && probs && rows > 0 && cols > 0 && rows < (1ll << 31), "fg_softmax_rows_f32_bf16: bad arguments");
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, scores, (bf16*)probs, cols,
                       scale);
    return fg_launch_status("fg_softmax_rows_f32_bf16");
}

int fg_vae_latent_to_cl_bf16(const void* z, const void* mean, const void* inv_std, void* out, int C, int T, int H, int W,
                             fg_stream_t stream) {
    FG_CHECK_ARG(z && mean && inv_std && out && C > 0 && T > 0 && H > 0 && W > 0, "fg_vae_latent_to_cl_bf16: bad arguments");
    const int64_t thw = (int64_t)T * H * W;
    hipLaunchKernelGGL(latent_to_cl_kernel, dim3(grid_for(thw * C)), dim3(256), 0, (hipStream_t)stream, (const bf16*)z,
                       (const bf16*)mean, (const bf16*)inv_std, (bf16*)out, C, thw);
    return fg_launch_status("fg_vae_latent_to_cl_bf16");
}

int fg_vae_unpatchify_bf16(const void* x, void* video, int T, int H, int W, int F, int t0, int do_clamp, fg_stream_t stream) {
    FG_CHECK_ARG(x && video && T > 0 && H > 0 && W > 0 && t0 >= 0 && t0 + T <= F, "fg_vae_unpatchify_bf16: bad arguments");
    const int64_t total = (int64_t)3 * T * 4 * H * W;
    hipLaunchKernelGGL(unpatchify_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (bf16*)video,
                       T, H, W, F, t0, do_clamp);
    return fg_launch_status("fg_vae_unpatchify_bf16");
}

int fg_vae_tile_accumulate_bf16(const void* tile, void* values, void* weight, int C, int F, int Hv, int Wv, int th, int tw,
                                int y0, int x0, int border_h, int border_w, int bound_bits, fg_stream_t stream) {
    FG_CHECK_ARG(tile && values && weight && C > 0, "fg_vae_tile_accumulate_bf16: null pointer");
    FG_CHECK_ARG(F > 0 && th > 0 && tw > 0 && y0 >= 0 && x0 >= 0 && y0 + th <= Hv && x0 + tw <= Wv && border_h >= 0 && border_w >= 0,
                 "fg_vae_tile_accumulate_bf16: tile does not fit the canvas");
    FG_CHECK_ARG(((bound_bits & 1) || border_h <= th) && ((bound_bits & 4) || border_w <= tw), "fg_vae_tile_accumulate_bf16: border wider than tile");
    hipLaunchKernelGGL(tile_accumulate_kernel, dim3(grid_for((int64_t)F * th * tw)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16*)tile, (bf16*)values, (bf16*)weight, C, F, Hv, Wv, th, tw, y0, x0, border_h, border_w, bound_bits);
    return fg_launch_status("fg_vae_tile_accumulate_bf16");
}

int fg_vae_tile_finalize_bf16(void* values, const void* weight, int C, int F, int Hv, int Wv, int do_clamp, fg_stream_t stream) {
    FG_CHECK_ARG(values && weight && C > 0 && F > 0 && Hv > 0 && Wv > 0, "fg_vae_tile_finalize_bf16: bad arguments");
    const int64_t fhw = (int64_t)F * Hv * Wv;
    hipLaunchKernelGGL(tile_finalize_kernel, dim3(grid_for(C * fhw)), dim3(256), 0, (hipStream_t)stream, (bf16*)values,
                       (const bf16*)weight, C, fhw, do_clamp);
    return fg_launch_status("fg_vae_tile_finalize_bf16");
}

int fg_vae_patchify_bf16(const void* video, void* out, int T, int H, int W, fg_stream_t stream) {
    FG_CHECK_ARG(video && out && T > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "fg_vae_patchify_bf16: bad arguments");
    hipLaunchKernelGGL(patchify_kernel, dim3(grid_for((int64_t)T * H * W * 4)), dim3(256), 0, (hipStream_t)stream, (const bf16*)video,
                       (bf16*)out, T, H / 2, W / 2);
    return fg_launch_status("fg_vae_patchify_bf16");
}

int fg_avgdown3d_add_bf16(const void* x, const void* main_path, void* out, int T, int H, int W, int Cin, int Cout, int ft, int fs,
                          fg_stream_t stream) {
    FG_CHECK_ARG(x && main_path && out, "fg_avgdown3d_add_bf16: null pointer");
    FG_CHECK_ARG(T > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && (ft == 1 || ft == 2) && (fs == 1 || fs == 2) && H % fs == 0 &&
                     W % fs == 0 && (Cin * ft * fs * fs) % Cout == 0,
                 "fg_avgdown3d_add_bf16: bad sizes");
    const int64_t total = (int64_t)((T + ft - 1) / ft) * (H / fs) * (W / fs) * Cout;
    hipLaunchKernelGGL(avgdown3d_add_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const bf16*)x,
                       (const bf16*)main_path, (bf16*)out, T, H, W, Cin, Cout, ft, fs);
    return fg_launch_status("fg_avgdown3d_add_bf16");
}

int fg_vae_latent_from_cl_bf16(const void* x, const void* mean, const void* inv_std, void* out, int Z, int Cx, int T, int H, int W,
                               fg_stream_t stream) {
    FG_CHECK_ARG(x && mean && inv_std && out && Z > 0 && Cx >= Z && T > 0 && H > 0 && W > 0, "fg_vae_latent_from_cl_bf16: bad arguments");
    const int64_t thw = (int64_t)T * H * W;
    hipLaunchKernelGGL(latent_from_cl_kernel, dim3(grid_for(thw * Z)), dim3(256), 0, (hipStream_t)stream, (const bf16*)x,
                       (const bf16*)mean, (const bf16*)inv_std, (bf16*)out, Z, Cx, thw);
    return fg_launch_status("fg_vae_latent_from_cl_bf16");
}

int fg_video_to_uint8(const void* video, void* out_u8, int F, int H, int W, fg_stream_t stream) {
    FG_CHECK_ARG(video && out_u8 && F > 0 && H > 0 && W > 0, "fg_video_to_uint8: bad arguments");
    const int64_t fhw = (int64_t)F * H * W;
    hipLaunchKernelGGL(video_to_uint8_kernel, dim3(grid_for(3 * fhw)), dim3(256), 0, (hipStream_t)stream, (const bf16*)video,
                       (uint8_t*)out_u8, fhw);
    return fg_launch_status("fg_video_to_uint8");
}

}  // extern "C"

// DiT token-side kernels: LayerNorm+modulate, LayerNorm affine, gate-residual (+ fused next norm),
// RMSNorm(+RoPE), activations, CFG+Euler.  All are HBM-bound: one wave owns one token row, keeps it in
// registers (C <= 4096 -> <= 8 x 16-byte vectors per lane), reduces with wave shuffles (no LDS, no
// barrier) and touches every byte exactly once with 16-byte coalesced accesses.
// Rounding points mirror the reference's bf16 tensor arithmetic op by op (see include/fairygen_hip.h).
#include "common.h"

// Row kernels (one wave per token row held in registers): second launch-bound = minimum waves per SIMD the register
// allocation must allow (FG_ROW_MIN_WAVES; measured in tools/microbench.py elementwise).
#ifndef FG_ROW_MIN_WAVES
#define FG_ROW_MIN_WAVES 4
#endif
#define FG_ROW_BOUNDS __launch_bounds__(256, FG_ROW_MIN_WAVES)

namespace {

constexpr int kMaxVec = 8;            // 8 lanes-vectors * 64 lanes * 8 elems = 4096 channels max
constexpr int kRowsPerBlock = 4;      // 4 waves per 256-thread workgroup

struct Row {
    float v[kMaxVec][8];
};

__device__ __forceinline__ void load_row(const bf16* p, int C, int lane, Row& r) {
    const int nvec = C >> 3;
#pragma unroll
    for (int i = 0; i < kMaxVec; ++i) {
        const int vi = lane + i * 64;
        if (vi < nvec) {
            bf16x8 t = *reinterpret_cast<const bf16x8*>(p + (int64_t)vi * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) r.v[i][j] = (float)t[j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) r.v[i][j] = 0.f;
        }
    }
}

__device__ __forceinline__ bf16x8 ld8(const bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ void st8(bf16* p, const float* f) {
    bf16x8 t;
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = (bf16)f[j];
    *reinterpret_cast<bf16x8*>(p) = t;
}

// mean / rstd of a row held in registers (two-pass, fp32).
__device__ __forceinline__ void row_moments(const Row& r, int C, int lane, float eps, float& mean, float& rstd) {
    const int nvec = C >> 3;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxVec; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) s += r.v[i][j];
    mean = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxVec; ++i) {
        if (lane + i * 64 < nvec) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = r.v[i][j] - mean;
                q += d * d;
            }
        }
    }
    const float var = wave_sum(q) / (float)C;
    rstd = 1.0f / sqrtf(var + eps);
}

__device__ __forceinline__ int64_t mod_row(int64_t row, int64_t mod_rows, int64_t first_rows) {
    return mod_rows == 1 ? 0 : (mod_rows == 2 ? (row < first_rows ? 0 : 1) : row);
}

// norm_out = modulate(LN(row)) or LN(row)*w+b, written from registers.
template <int MODE>
__device__ __forceinline__ void norm_store(const Row& r, int C, int lane, float eps, const bf16* p0, const bf16* p1,
                                           bf16* out) {
    const int nvec = C >> 3;
    float mean, rstd;
    row_moments(r, C, lane, eps, mean, rstd);
#pragma unroll
    for (int i = 0; i < kMaxVec; ++i) {
        const int vi = lane + i * 64;
        if (vi < nvec) {
            const bf16x8 a = ld8(p0 + (int64_t)vi * 8);
            const bf16x8 b = ld8(p1 + (int64_t)vi * 8);
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float n = (r.v[i][j] - mean) * rstd;
                if (MODE == 0) {  // p0 = shift, p1 = scale
                    const float y = rbf(n);
                    const float s1 = rbf(1.0f + (float)b[j]);
                    o[j] = rbf(y * s1) + (float)a[j];
                } else {          // p0 = weight, p1 = bias
                    o[j] = n * (float)a[j] + (float)b[j];
                }
            }
            st8(out + (int64_t)vi * 8, o);
        }
    }
}

// Two fp32 -> two e4m3 bytes in the low half of a dword (v_cvt_pk_fp8_f32: RNE, OCP e4m3fn on gfx950).
__device__ __forceinline__ uint32_t cvt2_fp8(float a, float b) {
    return (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false) & 0xffffu;
}

// The same normalised row as norm_store, but handed to the fp8 Linear that consumes it (AutoWrappedLinear.fp8_linear,
// core/vram/layers.py:331-342) without a trip through HBM: the bf16-rounded values stay in registers, the row maximum gives the
// dynamic scale, the row leaves as e4m3 bytes + one fp32 scale — exactly fg_fp8_quant_rows_bf16's arithmetic on norm_store's output.
template <int MODE>
__device__ __forceinline__ void norm_store_fp8(Row& r, int C, int lane, float eps, const bf16* p0, const bf16* p1,
                                               uint8_t* out8, float* scale_out, float fp8_max) {
    const int nvec = C >> 3;
    float mean, rstd;
    row_moments(r, C, lane, eps, mean, rstd);
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxVec; ++i) {
        const int vi = lane + i * 64;
        if (vi < nvec) {
            const bf16x8 a = ld8(p0 + (int64_t)vi * 8);
            const bf16x8 b = ld8(p1 + (int64_t)vi * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float n = (r.v[i][j] - mean) * rstd;
                float o;
                if (MODE == 0) {
                    const float y = rbf(n);
                    const float s1 = rbf(1.0f + (float)b[j]);
                    o = rbf(y * s1) + (float)a[j];
                } else {
                    o = n * (float)a[j] + (float)b[j];
                }
                r.v[i][j] = rbf(o);
                amax = fmaxf(amax, fabsf(r.v[i][j]));
            }
        }
    }
    amax = wave_max(amax);
    const float sc = fmaxf(rbf(amax / fp8_max), 1.0f);
    const float denom = sc + 1e-8f;
    if (lane == 0) *scale_out = sc;
#pragma unroll
    for (int i = 0; i < kMaxVec; ++i) {
        const int vi = lane + i * 64;
        if (vi < nvec) {
            u32x2 w;
            w[0] = cvt2_fp8(r.v[i][0] / denom, r.v[i][1] / denom) | (cvt2_fp8(r.v[i][2] / denom, r.v[i][3] / denom) << 16);
            w[1] = cvt2_fp8(r.v[i][4] / denom, r.v[i][5] / denom) | (cvt2_fp8(r.v[i][6] / denom, r.v[i][7] / denom) << 16);
            *reinterpret_cast<u32x2*>(out8 + (int64_t)vi * 8) = w;
        }
    }
}

__global__ FG_ROW_BOUNDS void ln_modulate_fp8_kernel(const bf16* __restrict__ x, const bf16* __restrict__ shift,
                                                     const bf16* __restrict__ scale, uint8_t* __restrict__ out8,
                                                     float* __restrict__ scale_out, int64_t rows, int C, float eps, int64_t mod_rows,
                                                     int64_t first_rows, int64_t mod_ld, float fp8_max) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);
    if (row >= rows) return;
    Row r;
    load_row(x + row * C, C, lane, r);
    const int64_t m = mod_row(row, mod_rows, first_rows);
    norm_store_fp8<0>(r, C, lane, eps, shift + m * mod_ld, scale + m * mod_ld, out8 + row * C, scale_out + row, fp8_max);
}

__global__ FG_ROW_BOUNDS void ln_modulate_kernel(const bf16* __restrict__ x, const bf16* __restrict__ shift,
                                                          const bf16* __restrict__ scale, bf16* __restrict__ out,
                                                          int64_t rows, int C, float eps, int64_t mod_rows,
                                                          int64_t first_rows, int64_t mod_ld) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);
    if (row >= rows) return;
    Row r;
    load_row(x + row * C, C, lane, r);
    const int64_t m = mod_row(row, mod_rows, first_rows);
    norm_store<0>(r, C, lane, eps, shift + m * mod_ld, scale + m * mod_ld, out + row * C);
}

__global__ FG_ROW_BOUNDS void ln_affine_kernel(const bf16* __restrict__ x, const bf16* __restrict__ w,
                                                        const bf16* __restrict__ b, bf16* __restrict__ out,
                                                        int64_t rows, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);
    if (row >= rows) return;
    Row r;
    load_row(x + row * C, C, lane, r);
    norm_store<1>(r, C, lane, eps, w, b, out + row * C);
}

// x_out = x + gate*y ; optional fused norm of x_out.  MODE: -1 none, 0 modulate, 1 affine.
template <int MODE>
__global__ FG_ROW_BOUNDS void residual_kernel(const bf16* __restrict__ x, const bf16* __restrict__ y,
                                                       const bf16* __restrict__ gate, bf16* __restrict__ x_out,
                                                       const bf16* __restrict__ p0, const bf16* __restrict__ p1,
                                                       bf16* __restrict__ norm_out, int64_t rows, int C, float eps,
                                                       int64_t mod_rows, int64_t first_rows, int64_t mod_ld,
                                                       uint8_t* __restrict__ norm_fp8 = nullptr, float* __restrict__ norm_scale = nullptr,
                                                       float fp8_max = 0.f) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nvec = C >> 3;
    const int64_t m = mod_row(row, mod_rows, first_rows);
    Row r;
#pragma unroll
    for (int i = 0; i < kMaxVec; ++i) {
        const int vi = lane + i * 64;
        if (vi < nvec) {
            const bf16x8 xv = ld8(x + row * C + (int64_t)vi * 8);
            const bf16x8 yv = ld8(y + row * C + (int64_t)vi * 8);
            if (gate != nullptr) {
                const bf16x8 gv = ld8(gate + m * mod_ld + (int64_t)vi * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) r.v[i][j] = rbf((float)xv[j] + rbf((float)gv[j] * (float)yv[j]));
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) r.v[i][j] = rbf((float)xv[j] + (float)yv[j]);
            }
            st8(x_out + row * C + (int64_t)vi * 8, r.v[i]);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) r.v[i][j] = 0.f;
        }
    }
    if (norm_fp8 != nullptr) {      // the norm feeds an fp8 Linear only: e4m3 row + scale instead of the bf16 row
        if (MODE == 0) norm_store_fp8<0>(r, C, lane, eps, p0 + m * mod_ld, p1 + m * mod_ld, norm_fp8 + row * C, norm_scale + row, fp8_max);
        if (MODE == 1) norm_store_fp8<1>(r, C, lane, eps, p0, p1, norm_fp8 + row * C, norm_scale + row, fp8_max);
        return;
    }
    if (MODE == 0) norm_store<0>(r, C, lane, eps, p0 + m * mod_ld, p1 + m * mod_ld, norm_out + row * C);
    if (MODE == 1) norm_store<1>(r, C, lane, eps, p0, p1, norm_out + row * C);
}

// RMSNorm over the whole row, * weight, then RoPE on adjacent pairs.  F32TAB false: fp64 cos / sin tables and an fp64
// rotation, the reference's arithmetic (43 us of fp64 VALU + 16-byte table loads per pair on top of the 77 us the
// HBM-bound norm takes at N = 27 280).  F32TAB true: `ct` is ONE interleaved fp32 table (rows, head_dim/2, {cos, sin}) =
// the fp64 table rounded once, rotation as two fp32 FMAs: the bf16 result differs from the fp64 one only where the
// exact value lies within ~2e-7 relative of a bf16 rounding boundary (measured in tests/test_hip_kernels.py).
// PLAIN: plain rows (group_cols == C) and head_dim a power of two — the runtime `/ group_cols` and `% head_dim` (an integer division
// each, per vector: about as many VALU instructions as the norm itself) become a constant and a mask.
template <bool F32TAB, bool PLAIN = false>
__global__ FG_ROW_BOUNDS void rmsnorm_rope_kernel(const bf16* __restrict__ x, int64_t ldx,
                                                           const bf16* __restrict__ w, const void* __restrict__ ctv,
                                                           const void* __restrict__ stv, bf16* __restrict__ out,
                                                           int64_t rows, int C, int head_dim, float eps, int group_cols,
                                                           int64_t out_group_stride, int64_t out_ld) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nvec = C >> 3;
    Row r;
    load_row(x + row * ldx, C, lane, r);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxVec; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) q += r.v[i][j] * r.v[i][j];
    const float ms = wave_sum(q) / (float)C;
    const float rinv = 1.0f / sqrtf(ms + eps);
    const int half = head_dim >> 1;
    // fp32 table: a lane's vectors are 512 channels apart, so when head_dim divides 512 (128 here) every vector of the lane
    // sits at the same channel offset inside its head and needs the SAME four (cos, sin) pairs: loaded once per row
    const bool hoist = F32TAB && ctv != nullptr && (512 % head_dim) == 0;
    f32x4 h0 = {0.f, 0.f, 0.f, 0.f}, h1 = {0.f, 0.f, 0.f, 0.f};
    if (hoist) {
        const int d_lane = PLAIN ? ((lane * 8) & (head_dim - 1)) : ((lane * 8) % head_dim);
        const f32x4* tp = reinterpret_cast<const f32x4*>(static_cast<const float*>(ctv) + (row * half + (d_lane >> 1)) * 2);
        h0 = tp[0];
        h1 = tp[1];
    }
#pragma unroll
    for (int i = 0; i < kMaxVec; ++i) {
        const int vi = lane + i * 64;
        if (vi < nvec) {
            const bf16x8 wv = ld8(w + (int64_t)vi * 8);
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = rbf(rbf(r.v[i][j] * rinv) * (float)wv[j]);
            if (ctv != nullptr) {
                const int d0 = PLAIN ? ((vi * 8) & (head_dim - 1)) : ((vi * 8) % head_dim);          // channel within the head, multiple of 8
                if (F32TAB) {
                    f32x4 t0 = h0, t1 = h1;                  // (c0,s0,c1,s1) (c2,s2,c3,s3)
                    if (!hoist) {
                        const f32x4* tp = reinterpret_cast<const f32x4*>(static_cast<const float*>(ctv) + (row * half + (d0 >> 1)) * 2);
                        t0 = tp[0];
                        t1 = tp[1];
                    }
                    const float cs[8] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3]};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float a = o[2 * j], b = o[2 * j + 1], c = cs[2 * j], sn = cs[2 * j + 1];
                        o[2 * j] = rbf(__builtin_fmaf(a, c, -(b * sn)));
                        o[2 * j + 1] = rbf(__builtin_fmaf(a, sn, b * c));
                    }
                } else {
                    const double* cp = static_cast<const double*>(ctv) + row * half + (d0 >> 1);
                    const double* sp = static_cast<const double*>(stv) + row * half + (d0 >> 1);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const double a = (double)o[2 * j], b = (double)o[2 * j + 1];
                        const double c = cp[j], sn = sp[j];
                        o[2 * j] = (float)(bf16)(a * c - b * sn);
                        o[2 * j + 1] = (float)(bf16)(a * sn + b * c);
                    }
                }
            }
            // column block g = col / group_cols goes to its own (rows, out_ld) plane (group_cols == C: plain rows)
            const int col = vi * 8, g = PLAIN ? 0 : col / group_cols;
            st8(out + g * out_group_stride + row * out_ld + (col - g * group_cols), o);
        }
    }
}

// dst[g][r][0..cols) = src[g][r][0..cols) with independent group / row strides on both sides (16-byte vectors):
// the head-group <-> token re-layouts around the Ulysses all-to-alls.
__global__ __launch_bounds__(256) void copy_groups_kernel(const bf16* __restrict__ src, int64_t sgs, int64_t sld,
                                                          bf16* __restrict__ dst, int64_t dgs, int64_t dld, int groups,
                                                          int64_t rows, int vec_per_row) {
    const int64_t total = (int64_t)groups * rows * vec_per_row;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t rg = i / vec_per_row;            // row-major over (row, group): a wave reads one source row
        const int v = (int)(i - rg * vec_per_row);
        const int64_t r = rg / groups;
        const int g = (int)(rg - r * groups);
        *reinterpret_cast<u32x4*>(dst + g * dgs + r * dld + v * 8) =
            *reinterpret_cast<const u32x4*>(src + g * sgs + r * sld + v * 8);
    }
}


template <int KIND>
__global__ __launch_bounds__(256) void act_kernel(const bf16* __restrict__ x, bf16* __restrict__ out, int64_t nvec) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
        const bf16x8 v = ld8(x + i * 8);
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = KIND == 0 ? silu_f((float)v[j]) : gelu_tanh_f((float)v[j]);
        st8(out + i * 8, o);
    }
}

__global__ __launch_bounds__(256) void cfg_euler_kernel(const bf16* __restrict__ lat, const bf16* __restrict__ posi,
                                                        const bf16* __restrict__ nega, bf16* __restrict__ out,
                                                        int64_t n, float cfg, float dsigma) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float p = (float)posi[i];
        float pred = p;
        if (nega != nullptr) {
            const float ng = (float)nega[i];
            pred = rbf(ng + rbf(cfg * rbf(p - ng)));
        }
        out[i] = (bf16)((float)lat[i] + rbf(pred * dsigma));
    }
}

inline int check_rows(const char* fn, int64_t rows, int C, int64_t mod_rows, int64_t first_rows) {
    FG_CHECK_ARG(rows >= 0 && C > 0 && (C % 8) == 0 && C <= kMaxVec * 512, "%s: need C %% 8 == 0 and C <= %d (got rows=%lld C=%d)",
                 fn, kMaxVec * 512, (long long)rows, C);
    FG_CHECK_ARG(mod_rows == 1 || mod_rows == 2 || mod_rows == rows, "%s: mod_rows must be 1, 2 or rows (got %lld)", fn,
                 (long long)mod_rows);
    FG_CHECK_ARG(first_rows >= 0 && first_rows <= rows, "%s: first_rows out of range", fn);
    return FG_OK;
}

inline dim3 row_grid(int64_t rows) { return dim3((unsigned)((rows + kRowsPerBlock - 1) / kRowsPerBlock)); }

}  // namespace

extern "C" {

int fg_ln_modulate_bf16(const void* x, const void* shift, const void* scale, void* out, int64_t rows, int C, float eps,
                        int64_t mod_rows, int64_t first_rows, int64_t mod_ld, fg_stream_t stream) {
    if (int e = check_rows("fg_ln_modulate_bf16", rows, C, mod_rows, first_rows)) return e;
    FG_CHECK_ARG(x && shift && scale && out, "fg_ln_modulate_bf16: null pointer");
    FG_CHECK_ARG(FG_ALIGNED16(x) && FG_ALIGNED16(shift) && FG_ALIGNED16(scale) && FG_ALIGNED16(out) && mod_ld % 8 == 0,
                 "fg_ln_modulate_bf16: pointers / mod_ld must be 16-byte aligned");
    if (rows == 0) return FG_OK;
    hipLaunchKernelGGL(ln_modulate_kernel, row_grid(rows), dim3(256), 0, (hipStream_t)stream, (const bf16*)x,
                       (const bf16*)shift, (const bf16*)scale, (bf16*)out, rows, C, eps, mod_rows, first_rows, mod_ld);
    return fg_launch_status("fg_ln_modulate_bf16");
}

int fg_ln_modulate_fp8_bf16(const void* x, const void* shift, const void* scale, void* out_fp8, float* out_scale, int64_t rows, int C,
                            float eps, int64_t mod_rows, int64_t first_rows, int64_t mod_ld, float fp8_max, fg_stream_t stream) {
    if (int e = check_rows("fg_ln_modulate_fp8_bf16", rows, C, mod_rows, first_rows)) return e;
    FG_CHECK_ARG(x && shift && scale && out_fp8 && out_scale, "fg_ln_modulate_fp8_bf16: null pointer");
    FG_CHECK_ARG(FG_ALIGNED16(x) && FG_ALIGNED16(shift) && FG_ALIGNED16(scale) && (((uintptr_t)out_fp8) & 7) == 0 && mod_ld % 8 == 0 &&
                     fp8_max > 0.f,
                 "fg_ln_modulate_fp8_bf16: pointers / mod_ld must be 16-byte aligned (out_fp8: 8), fp8_max positive");
    if (rows == 0) return FG_OK;
    hipLaunchKernelGGL(ln_modulate_fp8_kernel, row_grid(rows), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (const bf16*)shift,
                       (const bf16*)scale, (uint8_t*)out_fp8, out_scale, rows, C, eps, mod_rows, first_rows, mod_ld, fp8_max);
    return fg_launch_status("fg_ln_modulate_fp8_bf16");
}

int fg_residual_ln_fp8_bf16(const void* x, const void* y, const void* gate, void* x_out, const void* p0, const void* p1,
                            void* norm_fp8, float* norm_scale, int mode, int64_t rows, int C, float eps, int64_t mod_rows,
                            int64_t first_rows, int64_t mod_ld, float fp8_max, fg_stream_t stream) {
    FG_CHECK_ARG(mode == 0 || mode == 1, "fg_residual_ln_fp8_bf16: mode must be 0 (modulate) or 1 (affine)");
    const bool need_mod = gate != nullptr || mode == 0;
    if (int e = check_rows("fg_residual_ln_fp8_bf16", rows, C, need_mod ? mod_rows : 1, need_mod ? first_rows : 0)) return e;
    FG_CHECK_ARG(x && y && x_out && p0 && p1 && norm_fp8 && norm_scale, "fg_residual_ln_fp8_bf16: null pointer");
    FG_CHECK_ARG(FG_ALIGNED16(x) && FG_ALIGNED16(y) && FG_ALIGNED16(gate) && FG_ALIGNED16(x_out) && FG_ALIGNED16(p0) &&
                     FG_ALIGNED16(p1) && (((uintptr_t)norm_fp8) & 7) == 0 && mod_ld % 8 == 0 && fp8_max > 0.f,
                 "fg_residual_ln_fp8_bf16: pointers / mod_ld must be 16-byte aligned (norm_fp8: 8), fp8_max positive");
    if (rows == 0) return FG_OK;
    if (!need_mod) { mod_rows = 1; first_rows = 0; }
    if (mode == 0)
        hipLaunchKernelGGL(residual_kernel<0>, row_grid(rows), dim3(256), 0, (hipStream_t)stream, (const bf16*)x,
                           (const bf16*)y, (const bf16*)gate, (bf16*)x_out, (const bf16*)p0, (const bf16*)p1,
                           (bf16*)nullptr, rows, C, eps, mod_rows, first_rows, mod_ld, (uint8_t*)norm_fp8, norm_scale, fp8_max);
    else
        hipLaunchKernelGGL(residual_kernel<1>, row_grid(rows), dim3(256), 0, (hipStream_t)stream, (const bf16*)x,
                           (const bf16*)y, (const bf16*)gate, (bf16*)x_out, (const bf16*)p0, (const bf16*)p1,
                           (bf16*)nullptr, rows, C, eps, mod_rows, first_rows, mod_ld, (uint8_t*)norm_fp8, norm_scale, fp8_max);
    return fg_launch_status("fg_residual_ln_fp8_bf16");
}

int fg_ln_affine_bf16(const void* x, const void* w, const void* b, void* out, int64_t rows, int C, float eps,
                      fg_stream_t stream) {
    if (int e = check_rows("fg_ln_affine_bf16", rows, C, 1, 0)) return e;
    FG_CHECK_ARG(x && w && b && out, "fg_ln_affine_bf16: null pointer");
    FG_CHECK_ARG(FG_ALIGNED16(x) && FG_ALIGNED16(w) && FG_ALIGNED16(b) && FG_ALIGNED16(out),
                 "fg_ln_affine_bf16: pointers must be 16-byte aligned");
    if (rows == 0) return FG_OK;
    hipLaunchKernelGGL(ln_affine_kernel, row_grid(rows), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (const bf16*)w,
                       (const bf16*)b, (bf16*)out, rows, C, eps);
    return fg_launch_status("fg_ln_affine_bf16");
}

int fg_gate_residual_bf16(const void* x, const void* y, const void* gate, void* out, int64_t rows, int C, int64_t mod_rows,
                          int64_t first_rows, int64_t mod_ld, fg_stream_t stream) {
    if (int e = check_rows("fg_gate_residual_bf16", rows, C, gate ? mod_rows : 1, gate ? first_rows : 0)) return e;
    FG_CHECK_ARG(x && y && out, "fg_gate_residual_bf16: null pointer");
    FG_CHECK_ARG(FG_ALIGNED16(x) && FG_ALIGNED16(y) && FG_ALIGNED16(gate) && FG_ALIGNED16(out) && mod_ld % 8 == 0,
                 "fg_gate_residual_bf16: pointers / mod_ld must be 16-byte aligned");
    if (rows == 0) return FG_OK;
    hipLaunchKernelGGL(residual_kernel<-1>, row_grid(rows), dim3(256), 0, (hipStream_t)stream, (const bf16*)x,
                       (const bf16*)y, (const bf16*)gate, (bf16*)out, (const bf16*)nullptr, (const bf16*)nullptr,
                       (bf16*)nullptr, rows, C, 0.f, gate ? mod_rows : 1, gate ? first_rows : 0, mod_ld);
    return fg_launch_status("fg_gate_residual_bf16");
}

int fg_residual_ln_bf16(const void* x, const void* y, const void* gate, void* x_out, const void* p0, const void* p1,
                        void* norm_out, int mode, int64_t rows, int C, float eps, int64_t mod_rows, int64_t first_rows,
                        int64_t mod_ld, fg_stream_t stream) {
    FG_CHECK_ARG(mode == 0 || mode == 1, "fg_residual_ln_bf16: mode must be 0 (modulate) or 1 (affine)");
    const bool need_mod = gate != nullptr || mode == 0;
    if (int e = check_rows("fg_residual_ln_bf16", rows, C, need_mod ? mod_rows : 1, need_mod ? first_rows : 0)) return e;
    FG_CHECK_ARG(x && y && x_out && p0 && p1 && norm_out, "fg_residual_ln_bf16: null pointer");
    FG_CHECK_ARG(FG_ALIGNED16(x) && FG_ALIGNED16(y) && FG_ALIGNED16(gate) && FG_ALIGNED16(x_out) && FG_ALIGNED16(p0) &&
                     FG_ALIGNED16(p1) && FG_ALIGNED16(norm_out) && mod_ld % 8 == 0,
                 "fg_residual_ln_bf16: pointers / mod_ld must be 16-byte aligned");
    if (rows == 0) return FG_OK;
    if (!need_mod) { mod_rows = 1; first_rows = 0; }
    if (mode == 0)
        hipLaunchKernelGGL(residual_kernel<0>, row_grid(rows), dim3(256), 0, (hipStream_t)stream, (const bf16*)x,
                           (const bf16*)y, (const bf16*)gate, (bf16*)x_out, (const bf16*)p0, (const bf16*)p1,
                           (bf16*)norm_out, rows, C, eps, mod_rows, first_rows, mod_ld);
    else
        hipLaunchKernelGGL(residual_kernel<1>, row_grid(rows), dim3(256), 0, (hipStream_t)stream, (const bf16*)x,
                           (const bf16*)y, (const bf16*)gate, (bf16*)x_out, (const bf16*)p0, (const bf16*)p1,
                           (bf16*)norm_out, rows, C, eps, mod_rows, first_rows, mod_ld);
    return fg_launch_status("fg_residual_ln_bf16");
}

int fg_rmsnorm_rope_bf16(const void* x, int64_t ldx, const void* weight, const void* cos_tab, const void* sin_tab,
                         int table_f32, void* out, int64_t rows, int C, int num_heads, float eps, fg_stream_t stream) {
    if (int e = check_rows("fg_rmsnorm_rope_bf16", rows, C, 1, 0)) return e;
    FG_CHECK_ARG(x && weight && out, "fg_rmsnorm_rope_bf16: null pointer");
    FG_CHECK_ARG(table_f32 ? (cos_tab != nullptr && sin_tab == nullptr) : ((cos_tab == nullptr) == (sin_tab == nullptr)),
                 "fg_rmsnorm_rope_bf16: fp64 mode takes both tables or neither, fp32 mode ONE interleaved table in cos_tab");
    FG_CHECK_ARG(num_heads > 0 && C % num_heads == 0 && (C / num_heads) % 8 == 0,
                 "fg_rmsnorm_rope_bf16: head_dim must be a multiple of 8");
    FG_CHECK_ARG(ldx >= C && ldx % 8 == 0 && FG_ALIGNED16(x) && FG_ALIGNED16(weight) && FG_ALIGNED16(out) &&
                     FG_ALIGNED16(cos_tab) && FG_ALIGNED16(sin_tab),
                 "fg_rmsnorm_rope_bf16: pointers / ldx must be 16-byte aligned");
    if (rows == 0) return FG_OK;
    const int hd = C / num_heads;
    const bool plain = (hd & (hd - 1)) == 0;          // plain rows + power-of-two head_dim: the division-free instantiation
#define FG_ROPE_LAUNCH(TAB, PL)                                                                                                  \
    hipLaunchKernelGGL((rmsnorm_rope_kernel<TAB, PL>), row_grid(rows), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, ldx,    \
                       (const bf16*)weight, cos_tab, sin_tab, (bf16*)out, rows, C, hd, eps, C, (int64_t)0, (int64_t)C)
    if (table_f32) { if (plain) FG_ROPE_LAUNCH(true, true); else FG_ROPE_LAUNCH(true, false); }
    else { if (plain) FG_ROPE_LAUNCH(false, true); else FG_ROPE_LAUNCH(false, false); }
#undef FG_ROPE_LAUNCH
    return fg_launch_status("fg_rmsnorm_rope_bf16");
}

int fg_rmsnorm_rope_grouped_bf16(const void* x, int64_t ldx, const void* weight, const void* cos_tab,
                                 const void* sin_tab, int table_f32, void* out, int64_t rows, int C, int num_heads, float eps,
                                 int group_cols, int64_t out_group_stride, int64_t out_ld, fg_stream_t stream) {
    if (int e = check_rows("fg_rmsnorm_rope_grouped_bf16", rows, C, 1, 0)) return e;
    FG_CHECK_ARG(x && weight && out, "fg_rmsnorm_rope_grouped_bf16: null pointer");
    FG_CHECK_ARG(table_f32 ? (cos_tab != nullptr && sin_tab == nullptr) : ((cos_tab == nullptr) == (sin_tab == nullptr)),
                 "fg_rmsnorm_rope_grouped_bf16: fp64 mode takes both tables or neither, fp32 mode ONE interleaved table in cos_tab");
    FG_CHECK_ARG(num_heads > 0 && C % num_heads == 0 && (C / num_heads) % 8 == 0,
                 "fg_rmsnorm_rope_grouped_bf16: head_dim must be a multiple of 8");
    FG_CHECK_ARG(group_cols > 0 && group_cols % 8 == 0 && C % group_cols == 0 && out_ld >= group_cols && out_ld % 8 == 0 &&
                     out_group_stride % 8 == 0 && out_group_stride >= 0,
                 "fg_rmsnorm_rope_grouped_bf16: group_cols must divide C; group_cols, out_ld, out_group_stride multiples of 8");
    FG_CHECK_ARG(ldx >= C && ldx % 8 == 0 && FG_ALIGNED16(x) && FG_ALIGNED16(weight) && FG_ALIGNED16(out) &&
                     FG_ALIGNED16(cos_tab) && FG_ALIGNED16(sin_tab),
                 "fg_rmsnorm_rope_grouped_bf16: pointers / ldx must be 16-byte aligned");
    if (rows == 0) return FG_OK;
    if (table_f32)
        hipLaunchKernelGGL((rmsnorm_rope_kernel<true, false>), row_grid(rows), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, ldx,
                           (const bf16*)weight, cos_tab, sin_tab, (bf16*)out, rows, C, C / num_heads, eps, group_cols,
                           out_group_stride, out_ld);
    else
        hipLaunchKernelGGL((rmsnorm_rope_kernel<false, false>), row_grid(rows), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, ldx,
                           (const bf16*)weight, cos_tab, sin_tab, (bf16*)out, rows, C, C / num_heads, eps, group_cols,
                           out_group_stride, out_ld);
    return fg_launch_status("fg_rmsnorm_rope_grouped_bf16");
}

int fg_copy_groups_bf16(const void* src, int64_t src_group_stride, int64_t src_ld, void* dst, int64_t dst_group_stride,
                        int64_t dst_ld, int groups, int64_t rows, int cols, fg_stream_t stream) {
    FG_CHECK_ARG(src && dst, "fg_copy_groups_bf16: null pointer");
    FG_CHECK_ARG(groups > 0 && rows >= 0 && cols > 0 && cols % 8 == 0, "fg_copy_groups_bf16: cols must be a positive multiple of 8");
    FG_CHECK_ARG(src_group_stride % 8 == 0 && src_ld % 8 == 0 && dst_group_stride % 8 == 0 && dst_ld % 8 == 0 &&
                     src_ld >= cols && dst_ld >= cols && FG_ALIGNED16(src) && FG_ALIGNED16(dst),
                 "fg_copy_groups_bf16: strides must be multiples of 8 elements, rows at least cols wide, pointers 16-byte aligned");
    if (rows == 0) return FG_OK;
    const int64_t total = (int64_t)groups * rows * (cols / 8);
    const unsigned grid = (unsigned)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
    hipLaunchKernelGGL(copy_groups_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16*)src, src_group_stride,
                       src_ld, (bf16*)dst, dst_group_stride, dst_ld, groups, rows, cols / 8);
    return fg_launch_status("fg_copy_groups_bf16");
}

int fg_act_bf16(const void* x, void* out, int64_t n, int kind, fg_stream_t stream) {
    FG_CHECK_ARG(x && out && n >= 0 && n % 8 == 0, "fg_act_bf16: n must be a multiple of 8");
    FG_CHECK_ARG(kind == 0 || kind == 1, "fg_act_bf16: kind must be 0 (silu) or 1 (gelu_tanh)");
    FG_CHECK_ARG(FG_ALIGNED16(x) && FG_ALIGNED16(out), "fg_act_bf16: pointers must be 16-byte aligned");
    if (n == 0) return FG_OK;
    const int64_t nvec = n / 8;
    const unsigned grid = (unsigned)((nvec + 255) / 256 < 8192 ? (nvec + 255) / 256 : 8192);
    if (kind == 0)
        hipLaunchKernelGGL(act_kernel<0>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (bf16*)out, nvec);
    else
        hipLaunchKernelGGL(act_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (bf16*)out, nvec);
    return fg_launch_status("fg_act_bf16");
}

int fg_cfg_euler_bf16(const void* latents, const void* posi, const void* nega, void* out, int64_t n, float cfg_scale,
                      float dsigma, fg_stream_t stream) {
    FG_CHECK_ARG(latents && posi && out && n >= 0, "fg_cfg_euler_bf16: null pointer");
    if (n == 0) return FG_OK;
    const unsigned grid = (unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(cfg_euler_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16*)latents,
                       (const bf16*)posi, (const bf16*)nega, (bf16*)out, n, cfg_scale, dsigma);
    return fg_launch_status("fg_cfg_euler_bf16");
}

}  // extern "C"

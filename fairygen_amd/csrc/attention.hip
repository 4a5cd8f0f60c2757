// Non-causal flash attention forward, head_dim 128, bf16 in / bf16 out, fp32 online softmax, for gfx950.
//
// Replaces flash_attention()/AttentionModule of the reference DiT (models/wan_video_dit.py:27-60,113-120):
// self-attention over N = f*h*w video tokens (27 280 at 704x1280x121) and cross-attention over 512 text
// tokens.  Roofline: MFMA bf16 (4*Nq*Nkv*128 FLOP per head; intensity ~Nkv/2 FLOP/B).
//
// Structure (one workgroup = 8 waves = 256 query rows of one head, 2 waves per SIMD):
//   * each wave owns 32 query rows; Q^T fragments live in registers for the whole kernel (32 VGPRs);
//   * K/V tiles of 64 keys are staged global -> registers -> LDS, double buffered, loads for tile t+1
//     issued before the math of tile t (issue-early / write-late), one barrier per tile;
//   * S^T = K * Q^T with v_mfma_f32_32x32x16_bf16 ("swapped" product): every lane ends up with 32 scores
//     of ONE query row, so the row max / row sum are lane-local plus a single lane<->lane+32 exchange;
//   * the S^T accumulators, converted pairwise to bf16, are directly the B operand of
//     O^T += V^T * P^T (no LDS round trip for P); V^T fragments come from the row-major V tile through
//     ds_read_b64_tr_b16 (hardware transpose);
//   * K tile: 256-B rows, 16-B chunks XOR-swizzled by (row & 15) -> conflict-free ds_read_b128;
//     V tile: 8-row x 32-column sub-tiles of 512 B with a chunk XOR -> conflict-free transposed reads;
//   * workgroup -> (head, q-block) map is XCD-aware: the 8 XCDs each walk a contiguous range of heads, so
//     the 32 CUs sharing an L2 stream the SAME head's K/V at the same time.
#include "common.h"

namespace {

constexpr int kD = 128;      // head dim
constexpr int kBN = 64;      // keys per tile
constexpr int kWaves = 8;
constexpr int kBM = 32 * kWaves;
constexpr int kThreads = 64 * kWaves;
constexpr int kTileBytes = kBN * kD * 2;  // 16 KiB

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ int k_lds_off(int row, int chunk) { return row * 256 + ((chunk ^ (row & 15)) << 4); }
__device__ __forceinline__ int v_lds_off(int row, int chunk) {
    return 2048 * (row >> 3) + 512 * (chunk >> 2) + 64 * (row & 7) + 16 * ((chunk & 3) ^ ((row >> 2) & 3));
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// max / sum across the lane pair (l, l^32) that shares a query row.
__device__ __forceinline__ float pair_max(float x) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float pair_sum(float x) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

__global__ __launch_bounds__(kThreads, 2) void attn_fwd_kernel(
    const bf16* __restrict__ q, int64_t ldq, const bf16* __restrict__ k, int64_t ldk, const bf16* __restrict__ v,
    int64_t ldv, bf16* __restrict__ out, int64_t Nq, int64_t Nkv, int H, int nqb, int total_blocks, float scale_log2e) {
    __shared__ __attribute__((aligned(16))) char smem[4 * kTileBytes];  // K0 K1 V0 V1
    char* const k_lds = smem;
    char* const v_lds = smem + 2 * kTileBytes;

    // ---- XCD-aware block remap: blocks b and b+8 share an XCD; give each XCD a contiguous logical range.
    int logical;
    {
        const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
        const int qd = total_blocks >> 3, rm = total_blocks & 7;
        logical = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + slot;
    }
    const int qb = logical % nqb;
    const int bh = logical / nqb;      // b*H + h
    const int b = bh / H, h = bh % H;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;

    const bf16* qp = q + (int64_t)b * Nq * ldq + (int64_t)h * kD;
    const bf16* kp = k + (int64_t)b * Nkv * ldk + (int64_t)h * kD;
    const bf16* vp = v + (int64_t)b * Nkv * ldv + (int64_t)h * kD;
    bf16* op = out + (int64_t)b * Nq * ((int64_t)H * kD) + (int64_t)h * kD;

    // ---- Q^T fragments (B operand of S^T = K Q^T): lane (r,hh) holds Q[row r][16*ks + 8*hh + 0..7].
    const int64_t q_row0 = (int64_t)qb * kBM + wave * 32;
    int64_t my_q = q_row0 + r;
    const bool q_valid = my_q < Nq;
    if (!q_valid) my_q = Nq - 1;
    bf16x8 qf[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qp + my_q * ldq + ks * 16 + hh * 8);

    // ---- tile staging: thread t moves rows (t>>4) and (t>>4)+32, 16-byte chunk (t&15), of K and V.
    const int st_row = tid >> 4, st_chunk = tid & 15;
    const int nt = (int)((Nkv + kBN - 1) / kBN);
    u32x4 kreg[2], vreg[2];
    auto stage_load = [&](int t) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int64_t row = (int64_t)t * kBN + st_row + 32 * i;
            if (row >= Nkv) row = Nkv - 1;
            kreg[i] = *reinterpret_cast<const u32x4*>(kp + row * ldk + st_chunk * 8);
            vreg[i] = *reinterpret_cast<const u32x4*>(vp + row * ldv + st_chunk * 8);
        }
    };
    auto stage_write = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = st_row + 32 * i;
            *reinterpret_cast<u32x4*>(k_lds + buf * kTileBytes + k_lds_off(row, st_chunk)) = kreg[i];
            *reinterpret_cast<u32x4*>(v_lds + buf * kTileBytes + v_lds_off(row, st_chunk)) = vreg[i];
        }
    };

    f32x16 o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) o[i][j] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // per-lane pieces of the transposed-read address: group g = lane>>4, i = lane&15, q_ = i>>2, p_ = i&3
    const int tr_g1 = (lane >> 4) & 1, tr_q = (lane & 15) >> 2, tr_p = lane & 3;

    stage_load(0);
    stage_write(0);
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        const bool has_next = t + 1 < nt;
        if (has_next) stage_load(t + 1);

        // ---- S^T = K Q^T : two 32-key sub-tiles, 8 k-steps of 16 over d
        f32x16 s[2];
        const char* kb = k_lds + cur * kTileBytes;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
            for (int j = 0; j < 16; ++j) s[sub][j] = 0.f;
            const int krow = sub * 32 + r;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(kb + k_lds_off(krow, 2 * ks + hh));
                s[sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[ks], s[sub], 0, 0, 0);
            }
        }
        // s[sub][reg] = score(key = 64t + 32sub + (reg&3) + 8(reg>>2) + 4hh, query r)
        if ((int64_t)(t + 1) * kBN > Nkv) {  // ragged last tile (wave-uniform branch)
            const int64_t kbase = (int64_t)t * kBN + 4 * hh;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    if (kbase + 32 * sub + (j & 3) + 8 * (j >> 2) >= Nkv) s[sub][j] = -INFINITY;
        }

        // ---- online softmax in the log2 domain
        float mt = s[0][0];
#pragma unroll
        for (int j = 1; j < 16; ++j) mt = fmaxf(mt, s[0][j]);
#pragma unroll
        for (int j = 0; j < 16; ++j) mt = fmaxf(mt, s[1][j]);
        mt = pair_max(mt);
        const float m_new = fmaxf(m_run, mt);
        const float alpha = fast_exp2((m_run - m_new) * scale_log2e);
        const float mb = m_new * scale_log2e;
        m_run = m_new;
        float psum = 0.f;
        bf16x8 pf[4];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float p = fast_exp2(s[sub][j] * scale_log2e - mb);
                psum += p;
                pf[sub * 2 + (j >> 3)][j & 7] = (bf16)p;
            }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) o[i][j] *= alpha;

        // ---- O^T += V^T P^T : 4 d-blocks x 4 k-steps of 16 keys
        const char* vb = v_lds + cur * kTileBytes;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                const int chunk = 4 * db + 2 * tr_g1 + (tr_p >> 1);
                const int row0 = 16 * kk + 4 * hh + tr_q;
                const s16x4 t0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (lds_s16x4*)(vb + v_lds_off(row0, chunk) + 8 * (tr_p & 1)));
                const s16x4 t1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (lds_s16x4*)(vb + v_lds_off(row0 + 8, chunk) + 8 * (tr_p & 1)));
                union { s16x4 h2[2]; bf16x8 f; } a;
                a.h2[0] = t0;
                a.h2[1] = t1;
                o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.f, pf[kk], o[db], 0, 0, 0);
            }
        }

        if (has_next) stage_write(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: normalise and store O[query r][d = 32db + 8g + 4hh + 0..3]
    const float l_tot = pair_sum(l_run);
    const float inv = 1.0f / l_tot;
    if (q_valid) {
        bf16* orow = op + my_q * ((int64_t)H * kD);
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 w4;
#pragma unroll
                for (int j = 0; j < 4; ++j) w4[j] = (bf16)(o[db][4 * g + j] * inv);
                *reinterpret_cast<bf16x4*>(orow + 32 * db + 8 * g + 4 * hh) = w4;
            }
    }
}

}  // namespace

extern "C" int fg_attn_fwd_bf16(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                                void* out, int B, int64_t Nq, int64_t Nkv, int H, int D, float scale,
                                fg_stream_t stream) {
    FG_CHECK_ARG(q && k && v && out, "fg_attn_fwd_bf16: null pointer");
    FG_CHECK_ARG(D == kD, "fg_attn_fwd_bf16: only head_dim 128 is supported (got %d)", D);
    FG_CHECK_ARG(B > 0 && H > 0 && Nq > 0 && Nkv > 0, "fg_attn_fwd_bf16: B, H, Nq, Nkv must be positive");
    const int64_t hd = (int64_t)H * D;
    FG_CHECK_ARG(ldq >= hd && ldk >= hd && ldv >= hd && ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0,
                 "fg_attn_fwd_bf16: leading dimensions must be >= H*D and multiples of 8");
    FG_CHECK_ARG(FG_ALIGNED16(q) && FG_ALIGNED16(k) && FG_ALIGNED16(v) && FG_ALIGNED16(out),
                 "fg_attn_fwd_bf16: pointers must be 16-byte aligned");
    const int64_t nqb = (Nq + kBM - 1) / kBM;
    const int64_t total = nqb * B * H;
    FG_CHECK_ARG(total < (1ll << 30), "fg_attn_fwd_bf16: grid too large");
    const float scale_log2e = scale * 1.4426950408889634f;
    hipLaunchKernelGGL(attn_fwd_kernel, dim3((unsigned)total), dim3(kThreads), 0, (hipStream_t)stream, (const bf16*)q, ldq,
                       (const bf16*)k, ldk, (const bf16*)v, ldv, (bf16*)out, Nq, Nkv, H, (int)nqb, (int)total, scale_log2e);
    return fg_launch_status("fg_attn_fwd_bf16");
}

// Non-causal flash attention forward, head_dim 128, bf16 in / bf16 out, fp32 online softmax, for gfx950.
//
// Replaces flash_attention()/AttentionModule of the reference DiT (models/wan_video_dit.py:27-60,113-120):
// self-attention over N = f*h*w video tokens (27 280 at 704x1280x121) and cross-attention over 512 text
// tokens.  Roofline: MFMA bf16 (4*Nq*Nkv*128 FLOP per head; intensity ~Nkv/2 FLOP/B).
//
// Structure (one workgroup = 8 waves = 256 query rows of one head, 2 waves per SIMD):
//   * each wave owns 32 query rows; Q^T fragments live in registers for the whole kernel (32 VGPRs);
//   * K/V tiles of 64 keys are staged global -> registers -> LDS (issue-early / write-late), K two tiles ahead
//     and V one tile ahead, 2 LDS buffers each, ONE barrier per tile;
//   * software pipeline inside a wave: S^T(t+1) = K(t+1) Q^T is issued on the matrix pipe WHILE the VALU runs
//     the online softmax of tile t (two score tiles live, statically named so nothing goes to scratch), then
//     O^T += V(t)^T P(t)^T; the loop body is branch-free so MFMA / VALU / LDS instructions interleave;
//   * S^T = K Q^T with v_mfma_f32_32x32x16_bf16 ("swapped" product): every lane holds 32 scores of ONE query row,
//     so row max / row sum are lane-local plus one lane<->lane+32 exchange, and the S^T accumulators, converted
//     pairwise to bf16, are directly the B operand of O^T += V^T P^T (no LDS round trip for P);
//     V^T fragments come from the row-major V tile through ds_read_b64_tr_b16 (hardware transpose);
//   * the O^T accumulators are rescaled only when some row's running max grew by more than 2^kDeferLog2
//     (deferred rescale: P stays bounded by 2^kDeferLog2, exact up to bf16 rounding of P);
//   * K tile: 256-B rows, 16-B chunks XOR-swizzled by (row & 15) -> conflict-free ds_read_b128;
//     V tile: 8-row x 32-column sub-tiles of 512 B with a chunk XOR -> conflict-free transposed reads;
//     all LDS addresses are loop-invariant per-lane bases + immediates;
//   * workgroup -> (head, q-block) map is XCD-aware: the 8 XCDs each walk a contiguous range of heads, so
//     the 32 CUs sharing an L2 stream the SAME head's K/V at the same time.
#include "common.h"

#ifndef FG_EXP
#define FG_EXP 0      // timing-only experiments (wrong results): 1 no barrier, 2 no LDS reads, 3 no softmax VALU
#endif

namespace {

constexpr int kD = 128;      // head dim
constexpr int kBN = 64;      // keys per tile
constexpr int kBM = 256;     // query rows per workgroup
constexpr int kTileBytes = kBN * kD * 2;  // 16 KiB
constexpr float kDeferLog2 = 6.0f;        // rescale O only if a row max grows by > 2^6 (P <= 64)

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ int k_lds_off(int row, int chunk) { return row * 256 + ((chunk ^ (row & 15)) << 4); }
__device__ __forceinline__ int v_lds_off(int row, int chunk) {
    return 2048 * (row >> 3) + 512 * (chunk >> 2) + 64 * (row & 7) + 16 * ((chunk & 3) ^ ((row >> 2) & 3));
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
// One instruction; plain fmaxf() makes hipcc canonicalise every MFMA output first (3x the VALU work).
__device__ __forceinline__ float max3(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// max / sum across the lane pair (l, l^32) that shares a query row.
__device__ __forceinline__ float pair_max(float x) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float pair_sum(float x) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// WAVES waves per workgroup, each owning QB blocks of 32 query rows (WAVES*QB*32 == 256):
//   <8,1>: two waves per SIMD, 32 rows each (<= 256 VGPRs);  <4,2>: one wave per SIMD with the whole 512-register
//   file, 64 rows each -- every K / V^T fragment read from LDS feeds two MFMAs (half the LDS traffic per FLOP).
template <int WAVES, int QB>
__global__ __launch_bounds__(WAVES * 64, WAVES == 8 ? 2 : 1) void attn_fwd_kernel(
    const bf16* __restrict__ q, int64_t ldq, const bf16* __restrict__ k, int64_t ldk, const bf16* __restrict__ v,
    int64_t ldv, bf16* __restrict__ out, int64_t Nq, int64_t Nkv, int H, int nqb, int total_blocks, float scale_log2e) {
    static_assert(WAVES * QB * 32 == kBM, "workgroup covers 256 query rows");
    constexpr int THREADS = WAVES * 64;
    constexpr int NST = (kBN * 16) / THREADS;      // 16-byte chunks per thread per tile (2 or 4)
    constexpr int ST_ROWS = THREADS / 16;          // rows covered per staging pass
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
    int64_t my_q[QB];
    bool q_valid[QB];
    bf16x8 qf[QB][8];
#pragma unroll
    for (int qi = 0; qi < QB; ++qi) {
        my_q[qi] = (int64_t)qb * kBM + (wave * QB + qi) * 32 + r;
        q_valid[qi] = my_q[qi] < Nq;
        if (!q_valid[qi]) my_q[qi] = Nq - 1;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
            qf[qi][ks] = *reinterpret_cast<const bf16x8*>(qp + my_q[qi] * ldq + ks * 16 + hh * 8);
    }

    // ---- tile staging: thread t moves rows (t>>4) + ST_ROWS*i, 16-byte chunk (t&15), of K and V.
    const int st_row = tid >> 4, st_chunk = tid & 15;
    const int nt = (int)((Nkv + kBN - 1) / kBN);
    int st_k[NST], st_v[NST];
#pragma unroll
    for (int i = 0; i < NST; ++i) {
        st_k[i] = k_lds_off(st_row + ST_ROWS * i, st_chunk);
        st_v[i] = v_lds_off(st_row + ST_ROWS * i, st_chunk);
    }
    // Buffer descriptors (wave-uniform) bound the loads to the head's rows: rows >= Nkv (ragged last tile, prefetch
    // past the end) come back as zeros from the hardware range check -- no clamps, 32-bit offsets.
    const uint32_t k_bytes = (uint32_t)((Nkv - 1) * ldk * 2 + kD * 2), v_bytes = (uint32_t)((Nkv - 1) * ldv * 2 + kD * 2);
    const __amdgpu_buffer_rsrc_t k_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(kp), 0, k_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(vp), 0, v_bytes, 0x00020000);
    const uint32_t k_tile_stride = (uint32_t)(kBN * ldk * 2), v_tile_stride = (uint32_t)(kBN * ldv * 2);
    uint32_t k_off[NST], v_off[NST];
#pragma unroll
    for (int i = 0; i < NST; ++i) {
        k_off[i] = (uint32_t)((st_row + ST_ROWS * i) * ldk * 2 + st_chunk * 16);
        v_off[i] = (uint32_t)((st_row + ST_ROWS * i) * ldv * 2 + st_chunk * 16);
    }
    u32x4 kreg[NST], vreg[NST];
    auto load_k = [&](int t) {
        const uint32_t base = (uint32_t)t * k_tile_stride;
#pragma unroll
        for (int i = 0; i < NST; ++i) kreg[i] = __builtin_amdgcn_raw_buffer_load_b128(k_rsrc, k_off[i] + base, 0, 0);
    };
    auto load_v = [&](int t) {
        const uint32_t base = (uint32_t)t * v_tile_stride;
#pragma unroll
        for (int i = 0; i < NST; ++i) vreg[i] = __builtin_amdgcn_raw_buffer_load_b128(v_rsrc, v_off[i] + base, 0, 0);
    };
    auto write_k = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NST; ++i) *reinterpret_cast<u32x4*>(k_lds + buf * kTileBytes + st_k[i]) = kreg[i];
    };
    auto write_v = [&](int buf) {
#pragma unroll
        for (int i = 0; i < NST; ++i) *reinterpret_cast<u32x4*>(v_lds + buf * kTileBytes + st_v[i]) = vreg[i];
    };

    // ---- loop-invariant per-lane LDS read addresses
    int k_rd[8];      // K fragment of k-step ks, sub-tile 0 (sub-tile 1: + 32 rows = + 8192 B)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) k_rd[ks] = k_lds_off(r, 2 * ks + hh);
    // transposed V reads: group g = lane>>4 (g>>1 == hh), i = lane&15, q_ = i>>2, p_ = i&3; block rows 16kk + 8u + 4hh + q_,
    // chunk 4db + 2(g&1) + (p_>>1): offset = base_u + 4096*kk + 512*db  (u = 0/1: second read is 8 rows further)
    const int tr_g1 = (lane >> 4) & 1, tr_q = (lane & 15) >> 2, tr_p = lane & 3;
    const int v_rd0 = v_lds_off(4 * hh + tr_q, 2 * tr_g1 + (tr_p >> 1)) + 8 * (tr_p & 1);
    const int v_rd1 = v_lds_off(8 + 4 * hh + tr_q, 2 * tr_g1 + (tr_p >> 1)) + 8 * (tr_p & 1);

    f32x16 o[QB][4];
    float m_run[QB], l_run[QB];
#pragma unroll
    for (int qi = 0; qi < QB; ++qi) {
        m_run[qi] = -INFINITY;
        l_run[qi] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) o[qi][i][j] = 0.f;
    }

    // s[qi][sub][reg] = score(key = 64t + 32sub + (reg&3) + 8(reg>>2) + 4hh, query block qi row r)
    auto qk_tile = [&](f32x16 (&s)[QB][2], int buf) {
        const char* kb = k_lds + buf * kTileBytes;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
            for (int qi = 0; qi < QB; ++qi)
#pragma unroll
                for (int j = 0; j < 16; ++j) s[qi][sub][j] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
#if FG_EXP == 2
                bf16x8 a = qf[0][(ks + 1) & 7];
                asm volatile("" : "+v"(a));
#else
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(kb + k_rd[ks] + sub * 8192);
#endif
#pragma unroll
                for (int qi = 0; qi < QB; ++qi)
                    s[qi][sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qf[qi][ks], s[qi][sub], 0, 0, 0);
            }
        }
    };
    auto mask_tile = [&](f32x16 (&s)[QB][2], int t) {
        const int64_t kbase = (int64_t)t * kBN + 4 * hh;
#pragma unroll
        for (int qi = 0; qi < QB; ++qi)
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    if (kbase + 32 * sub + (j & 3) + 8 * (j >> 2) >= Nkv) s[qi][sub][j] = -INFINITY;
    };

    // One pipeline step for tile t: sc = scores(t) (ready), sn <- scores(t+1) while softmax(sc) runs, then PV(t).
    auto step = [&](f32x16 (&sc)[QB][2], f32x16 (&sn)[QB][2], int t) {
        const int cur = t & 1;
        load_k(t + 2);
        load_v(t + 1);

        // ---- S^T(t+1) on the matrix pipe ...
        qk_tile(sn, cur ^ 1);
        // ---- ... under the online softmax of tile t (log2 domain) on the VALU
        bf16x8 pf[QB][4];
#pragma unroll
        for (int qi = 0; qi < QB; ++qi) {
            float mt = max3(sc[qi][0][0], sc[qi][1][0], sc[qi][0][1]);
            mt = max3(mt, sc[qi][1][1], sc[qi][0][2]);
#pragma unroll
            for (int j = 2; j < 15; ++j) mt = max3(mt, sc[qi][1][j], sc[qi][0][j + 1]);
            mt = pair_max(fmaxf(mt, sc[qi][1][15]));
            // deferred rescale: keep the stale max while no row of the wave outgrew it by more than 2^kDeferLog2
            const bool grow = (mt - m_run[qi]) * scale_log2e > kDeferLog2;      // true on the first tile (m_run = -inf)
            if (__builtin_amdgcn_ballot_w64(grow) != 0) {
                const float m_new = fmaxf(m_run[qi], mt);
                const float alpha = fast_exp2((m_run[qi] - m_new) * scale_log2e);
                m_run[qi] = m_new;
                l_run[qi] *= alpha;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 16; ++j) o[qi][i][j] *= alpha;
            }
            const float mb = m_run[qi] * scale_log2e;
            float psum = 0.f;
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int j = 0; j < 16; ++j) {
#if FG_EXP == 3
                    float p = sc[qi][sub][j];
                    asm volatile("" : "+v"(p));
#else
                    const float p = fast_exp2(sc[qi][sub][j] * scale_log2e - mb);
#endif
                    psum += p;
                    pf[qi][sub * 2 + (j >> 3)][j & 7] = (bf16)p;
                }
            l_run[qi] += psum;
        }

        // ---- O^T += V(t)^T P^T : 4 d-blocks x 4 k-steps of 16 keys
        const char* vb = v_lds + cur * kTileBytes;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                union { s16x4 h2[2]; bf16x8 f; } a;
#if FG_EXP == 2
                a.f = qf[0][(kk + db) & 7];
                asm volatile("" : "+v"(a.f));
#else
                a.h2[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vb + v_rd0 + 4096 * kk + 512 * db));
                a.h2[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vb + v_rd1 + 4096 * kk + 512 * db));
#endif
#pragma unroll
                for (int qi = 0; qi < QB; ++qi)
                    o[qi][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.f, pf[qi][kk], o[qi][db], 0, 0, 0);
            }
        }

        // K(t+2) replaces K(t) (last read in step t-1), V(t+1) replaces V(t-1) (last read in step t-1)
        write_k(cur);
        write_v(cur ^ 1);
        if ((int64_t)(t + 2) * kBN > Nkv && t + 1 < nt) mask_tile(sn, t + 1);      // ragged last tile (wave-uniform)
#if FG_EXP != 1
        __syncthreads();
#endif
    };

    // ---- prologue: K(0), V(0), K(1) resident; scores(0) computed
    f32x16 sA[QB][2], sB[QB][2];
    load_k(0);
    load_v(0);
    write_k(0);
    write_v(0);
    load_k(1);
    write_k(1);
    __syncthreads();
    qk_tile(sA, 0);
    if ((int64_t)kBN > Nkv) mask_tile(sA, 0);
    __syncthreads();      // every wave is done with K(0) before step 0 overwrites it with K(2)

    int t = 0;
    for (; t + 1 < nt; t += 2) {
        step(sA, sB, t);
        step(sB, sA, t + 1);
    }
    if (t < nt) step(sA, sB, t);

    // ---- epilogue: normalise and store O[query r][d = 32db + 8g + 4hh + 0..3]
#pragma unroll
    for (int qi = 0; qi < QB; ++qi) {
        const float inv = 1.0f / pair_sum(l_run[qi]);
        if (q_valid[qi]) {
            bf16* orow = op + my_q[qi] * ((int64_t)H * kD);
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    bf16x4 w4;
#pragma unroll
                    for (int j = 0; j < 4; ++j) w4[j] = (bf16)(o[qi][db][4 * g + j] * inv);
                    *reinterpret_cast<bf16x4*>(orow + 32 * db + 8 * g + 4 * hh) = w4;
                }
        }
    }
}

}  // namespace

#ifndef FG_ATTN_VARIANT
#define FG_ATTN_VARIANT 8      // 8: <8 waves, 1 q-block>; 4: <4 waves, 2 q-blocks>
#endif

extern "C" int fg_attn_fwd_bf16(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                                void* out, int B, int64_t Nq, int64_t Nkv, int H, int D, float scale,
                                fg_stream_t stream) {
    FG_CHECK_ARG(q && k && v && out, "fg_attn_fwd_bf16: null pointer");
    FG_CHECK_ARG(D == kD, "fg_attn_fwd_bf16: only head_dim 128 is supported (got %d)", D);
    FG_CHECK_ARG(B > 0 && H > 0 && Nq > 0 && Nkv > 0, "fg_attn_fwd_bf16: B, H, Nq, Nkv must be positive");
    FG_CHECK_ARG(scale > 0.f, "fg_attn_fwd_bf16: scale must be positive");
    const int64_t hd = (int64_t)H * D;
    FG_CHECK_ARG(ldq >= hd && ldk >= hd && ldv >= hd && ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0,
                 "fg_attn_fwd_bf16: leading dimensions must be >= H*D and multiples of 8");
    FG_CHECK_ARG(FG_ALIGNED16(q) && FG_ALIGNED16(k) && FG_ALIGNED16(v) && FG_ALIGNED16(out),
                 "fg_attn_fwd_bf16: pointers must be 16-byte aligned");
    const int64_t nqb = (Nq + kBM - 1) / kBM;
    const int64_t total = nqb * B * H;
    FG_CHECK_ARG(total < (1ll << 30), "fg_attn_fwd_bf16: grid too large");
    FG_CHECK_ARG((Nkv + 2 * kBN) * ldk * 2 < (1ll << 32) && (Nkv + 2 * kBN) * ldv * 2 < (1ll << 32),
                 "fg_attn_fwd_bf16: K/V of one batch element must span < 4 GiB (32-bit buffer offsets)");
    const float scale_log2e = scale * 1.4426950408889634f;
#if FG_ATTN_VARIANT == 4
    hipLaunchKernelGGL((attn_fwd_kernel<4, 2>), dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, (const bf16*)q, ldq,
                       (const bf16*)k, ldk, (const bf16*)v, ldv, (bf16*)out, Nq, Nkv, H, (int)nqb, (int)total, scale_log2e);
#else
    hipLaunchKernelGGL((attn_fwd_kernel<8, 1>), dim3((unsigned)total), dim3(512), 0, (hipStream_t)stream, (const bf16*)q, ldq,
                       (const bf16*)k, ldk, (const bf16*)v, ldv, (bf16*)out, Nq, Nkv, H, (int)nqb, (int)total, scale_log2e);
#endif
    return fg_launch_status("fg_attn_fwd_bf16");
}

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
#include <math.h>
#include <queue>
#include <type_traits>
#include <vector>

#ifndef FG_EXP
#define FG_EXP 0      // timing-only experiments (wrong results): 1 no barrier, 2 no LDS reads, 3 no softmax VALU
#endif

namespace {

constexpr int kD = 128;      // head dim
constexpr int kBN = 64;      // keys per tile
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

// RNE pack of two fp32 into one dword of bf16 (same rounding as the (bf16) cast), kept a single instruction so that it can
// be placed in a chosen MFMA gap.
__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
#ifndef FG_EXP_QK
#define FG_EXP_QK 21      // exponentials placed under the 16 QK gaps (the rest: one per PV gap from gap 16)
#endif
#ifndef FG_KDIST
#define FG_KDIST 3        // K fragment ds_read issued this many gaps before its MFMA
#endif
#ifndef FG_VDIST
#define FG_VDIST 2        // V^T fragment reads issued this many gaps before their MFMA
#endif
// Gap schedule of the 32 exponentials of a tile: FG_EXP_QK under the 16 QK gaps, then one per PV gap.
__host__ __device__ constexpr int exps_before(int g) {
    return g < 16 ? (FG_EXP_QK * g) / 16 : (FG_EXP_QK + (g - 16) < 32 ? FG_EXP_QK + (g - 16) : 32);
}
static_assert(exps_before(27) == 32, "the last exponential must be issued by gap 26 (its pack feeds the MFMA of gap 28)");

#ifndef FG_PK
#define FG_PK 0           // 1: v_pk_fma_f32 / v_pk_add_f32 on element pairs (two exponent arguments / two row-sum terms per issue).
#endif                    // Measured SLOWER (1155 -> 1067 TFLOP/s): tools/probes/valu_rate.hip shows a wave issues one VALU op per
                          // ~5.4 clocks whether it is packed or not and v_exp_f32 per 8.5, and the pairs tie the schedule down.
typedef float f32x2 __attribute__((ext_vector_type(2)));
// Packed fp32 ops as opaque single instructions (placed in a chosen MFMA gap like the other asm helpers).
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) {
    f32x2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

#ifndef FG_DOT2
#define FG_DOT2 0         // 1: row sum from the packed bf16 P words with v_dot2c_f32_bf16 (16 instead of 32 VALU ops per tile; the sum
#endif                    // is then over the bf16-rounded probabilities, the same values the PV product uses)
__device__ __forceinline__ float dot2c_bf16(uint32_t a, uint32_t b, float acc) {
    asm("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(acc) : "v"(a), "v"(b));
    return acc;
}

__device__ __forceinline__ float add_f32(float a, float b) {
    float r;
    asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
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
// Work decomposition (host-chosen, see choose_split): per (batch, head) the first `nfull` q-blocks are "direct" workgroups
// over the whole KV range; the last `R` q-blocks are each cut into `S` KV ranges ("pieces") whose un-normalised
// partial O / running max / row sum go to the workspace and are merged by attn_combine_kernel.  Direct workgroups get
// the low block ids (dispatched first), the small pieces fill the tail: wave quantisation (2568 workgroups on 256 CUs
// = 10.03 rounds at N = 27 280; 336 = 1.31 rounds for a 1/8 token shard) costs a fraction of a piece instead of a round.
#ifndef FG_ATTN_SCHED
#define FG_ATTN_SCHED 1       // 1: hand-placed per-MFMA-gap instruction schedule (step_sched); 0: compiler-scheduled step
#endif
#ifndef FG_ATTN_VARIANT
#define FG_ATTN_VARIANT 8      // 8: <8 waves, 1 q-block> (256 rows / WG); 4: <4 waves, 1 q-block> (128 rows, 2 WGs per CU)
#endif
#if FG_ATTN_VARIANT == 4
constexpr int kBM = 128;
#else
constexpr int kBM = 256;       // query rows per workgroup
#endif

struct AttnParams {
    const bf16* q; const bf16* k; const bf16* v; bf16* out;
    int64_t ldq, ldk, ldv, Nq, Nkv;
    int H, nqb, nfull, R, S, total_direct;
    float scale_log2e;
    float* o_part;      // [piece][256][128] fp32
    float* ml_part;     // [piece][256][2]   (running max in raw score units, row sum)
};

// SHORT_KV only names the instantiation: launches over <= 1024 keys (cross-attention against the text context) get their own
// kernel symbol, so a rocprofv3 --stats row averages self-attention launches alone and can be laid next to bench.py's
// `roofline.avg_launch_ms`.
template <int WAVES, int QB, bool SHORT_KV = false>
__global__ __launch_bounds__(WAVES * 64, QB == 1 ? 2 : 1) void attn_fwd_kernel(const AttnParams P) {
    const bf16* __restrict__ q = P.q; const bf16* __restrict__ k = P.k; const bf16* __restrict__ v = P.v;
    bf16* __restrict__ out = P.out;
    const int64_t ldq = P.ldq, ldk = P.ldk, ldv = P.ldv, Nq = P.Nq, Nkv = P.Nkv;
    const int H = P.H;
    const float scale_log2e = P.scale_log2e;
    static_assert(WAVES * QB * 32 == kBM, "workgroup rows");
    constexpr int THREADS = WAVES * 64;
    constexpr int NST = (kBN * 16) / THREADS;      // 16-byte chunks per thread per tile (2 or 4)
    constexpr int ST_ROWS = THREADS / 16;          // rows covered per staging pass
    __shared__ __attribute__((aligned(16))) char smem[4 * kTileBytes];  // K0 K1 V0 V1
    char* const k_lds = smem;
    char* const v_lds = smem + 2 * kTileBytes;

    const int nt = (int)((Nkv + kBN - 1) / kBN);
    int qb, bh, t_begin = 0, t_end = nt, piece = -1;
    if ((int)blockIdx.x < P.total_direct) {
        // XCD-aware remap: blocks b and b+8 share an XCD; give each XCD a contiguous logical range (of heads).
        const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
        const int qd = P.total_direct >> 3, rm = P.total_direct & 7;
        const int logical = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + slot;
        qb = logical % P.nfull;
        bh = logical / P.nfull;      // b*H + h
    } else {
        piece = (int)blockIdx.x - P.total_direct;
        const int sp = piece % P.S, rq = (piece / P.S) % P.R;
        bh = piece / (P.S * P.R);
        qb = P.nfull + rq;
        t_begin = (int)((int64_t)sp * nt / P.S);
        t_end = (int)((int64_t)(sp + 1) * nt / P.S);
    }
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
    int st_k[NST], st_v[NST];
#pragma unroll
    for (int i = 0; i < NST; ++i) {
        st_k[i] = k_lds_off(st_row + ST_ROWS * i, st_chunk);
        st_v[i] = v_lds_off(st_row + ST_ROWS * i, st_chunk);
    }
    // Buffer descriptors (wave-uniform) bound the loads to the head's rows: rows >= Nkv (ragged last tile, prefetch
    // past the end) come back as zeros from the hardware range check -- no clamps, 32-bit offsets.  The tile base stays in
    // the VGPR offset: the range check covers voffset + inst_offset only, not the scalar soffset.
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

#if FG_ATTN_SCHED
    static_assert(QB == 1, "the gap schedule is written for one 32-row q-block per wave");
    // Row max of a finished score tile (raw units), combined across the lane pair.
    auto tile_max = [&](const f32x16 (&s)[QB][2]) {
        float mt = max3(s[0][0][0], s[0][1][0], s[0][0][1]);
#pragma unroll
        for (int j = 1; j < 15; ++j) mt = max3(mt, s[0][1][j], s[0][0][j + 1]);
        return pair_max(fmaxf(mt, s[0][1][15]));
    };
    // Deferred rescale: keep the stale max while no row of the wave outgrew it by more than 2^kDeferLog2.
    auto absorb_max = [&](float mt) {
        const bool grow = (mt - m_run[0]) * scale_log2e > kDeferLog2;
        if (__builtin_amdgcn_ballot_w64(grow) != 0) {
            const float m_new = fmaxf(m_run[0], mt);
            const float alpha = fast_exp2((m_run[0] - m_new) * scale_log2e);
            m_run[0] = m_new;
            l_run[0] *= alpha;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 16; ++j) o[0][i][j] *= alpha;
        }
    };

    // One pipeline step for tile t, written as 32 MFMA "gaps" (16 of S^T(t+1) = K(t+1) Q^T, then 16 of O^T += V(t)^T P(t)^T)
    // with the other work of the step dealt out between them in program order and pinned there by sched_barrier:
    //   * exp2 / bf16 pack of tile t (its max is already folded into m_run: done at the end of the previous step):
    //     21 of the 32 exponentials under the QK gaps, 11 under the first 11 PV gaps, chunk kk complete before PV k-step kk;
    //   * K fragment reads FG_KDIST gaps ahead of their MFMA, V^T fragment reads FG_VDIST gaps ahead;
    //   * the row max of tile t+1 (scores complete after gap 15) under PV gaps 18..31; rescale check after the last gap;
    //   * the LDS writes of K(t+2) / V(t+1) (global loads issued at the top of the step) in the last 4 gaps.
    // kTail: the next tile is ragged or absent (last steps of a range): mask it, or skip its max, outside the gap stream.
    // cur_tag: the K/V ring parity as a compile-time constant (0 / 1: every LDS address is a loop-invariant per-lane base plus
    // an immediate) or -1 (runtime parity, tail steps).
    auto step_sched = [&](f32x16 (&sc)[QB][2], f32x16 (&sn)[QB][2], int t, auto tail_tag, auto cur_tag) {
        constexpr bool kTail = decltype(tail_tag)::value;
        constexpr int kCur = decltype(cur_tag)::value;
        const int cur = kCur >= 0 ? kCur : ((t - t_begin) & 1);
        const char* kb = k_lds + (cur ^ 1) * kTileBytes;
        const char* vb = v_lds + cur * kTileBytes;
        load_k(t + 2);
        load_v(t + 1);
        const float mb = m_run[0] * scale_log2e;
        float ps0 = 0.f, ps1 = 0.f, mt = 0.f;
        float pe[32];
        union PF { uint32_t w[4]; bf16x8 f; } pf[4];
        bf16x8 ka[16];
        union VF { s16x4 h2[2]; bf16x8 f; } va[16];
        auto read_k = [&](int g) { ka[g] = *reinterpret_cast<const bf16x8*>(kb + k_rd[g & 7] + (g >> 3) * 8192); };
        auto read_v = [&](int p) {
            va[p].h2[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vb + v_rd0 + 4096 * (p >> 2) + 512 * (p & 3)));
            va[p].h2[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vb + v_rd1 + 4096 * (p >> 2) + 512 * (p & 3)));
        };
        // exponential e (0..31) = element (sub = e >> 4, j = e & 15); pair unit u = e >> 1 -> word (u & 3) of pf[u >> 2]
#if FG_PK
        // Pairs (2u, 2u+1) are adjacent accumulator registers: one v_pk_fma_f32 makes both exponent arguments (issued with
        // the first exponential of the pair), one v_pk_add_f32 adds both terms to a two-lane row sum (issued with the pack).
        const f32x2 scale2 = {scale_log2e, scale_log2e}, nmb2 = {-mb, -mb};
        f32x2 ea[16], ps2 = {0.f, 0.f};
        // the packed FMA of a pair is issued one gap BEFORE the gap of its first exponential (no dependent issue stall)
        auto do_fma = [&](int u) {
            const int sub = u >> 3, j = (2 * u) & 15;
            ea[u] = pk_fma(f32x2{sc[0][sub][j], sc[0][sub][j + 1]}, scale2, nmb2);
        };
        auto do_exp = [&](int e) { pe[e] = fast_exp2((e & 1) ? ea[e >> 1].y : ea[e >> 1].x); };
        auto do_add = [&](int e) {};
        auto do_pack = [&](int u) {
            pf[u >> 2].w[u & 3] = cvt_pk_bf16(pe[2 * u], pe[2 * u + 1]);
            ps2 = pk_add(ps2, f32x2{pe[2 * u], pe[2 * u + 1]});
        };
#else
        auto do_exp = [&](int e) {
            const float p = fast_exp2(sc[0][e >> 4][e & 15] * scale_log2e - mb);
            pe[e] = p;
        };
        // Row-sum add as opaque asm (no SLP packing / sinking to the end of the step).  Issued one gap AFTER the exponential:
        // gfx950 needs a wait state between a transcendental and a VALU reader of its result, and the compiler's hazard
        // pass does not look inside inline asm.
#if FG_DOT2
        // the packed word of pair u is summed one pack later (no dependent issue right behind the conversion)
        const uint32_t ones = 0x3f803f80u;
        auto do_add = [&](int e) {};
        auto do_pack = [&](int u) {
            pf[u >> 2].w[u & 3] = cvt_pk_bf16(pe[2 * u], pe[2 * u + 1]);
            if (u > 0) {
                if (u & 1) ps1 = dot2c_bf16(pf[(u - 1) >> 2].w[(u - 1) & 3], ones, ps1);
                else ps0 = dot2c_bf16(pf[(u - 1) >> 2].w[(u - 1) & 3], ones, ps0);
            }
        };
#else
        auto do_add = [&](int e) { if (e & 1) ps1 = add_f32(ps1, pe[e]); else ps0 = add_f32(ps0, pe[e]); };
        auto do_pack = [&](int u) { pf[u >> 2].w[u & 3] = cvt_pk_bf16(pe[2 * u], pe[2 * u + 1]); };
#endif
#endif

        static_for<0, FG_KDIST>([&](auto i) { read_k(decltype(i)::value); });
#if FG_PK
        static_for<0, (exps_before(1) + 1) / 2>([&](auto i) { do_fma(decltype(i)::value); });
#endif
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, 32>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            // exponentials issued before gap g: exps_before(g); this gap issues [e0, e1)
            constexpr int e0 = exps_before(g), e1 = exps_before(g + 1), ep = g == 0 ? 0 : exps_before(g - 1);
            if constexpr (g < 16) {
                constexpr int sub = g >> 3, ks = g & 7;
                if constexpr (ks == 0) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) sn[0][sub][j] = 0.f;
                }
                sn[0][sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka[g], qf[0][ks], sn[0][sub], 0, 0, 0);
                if constexpr (g + FG_KDIST < 16) read_k(g + FG_KDIST);
            } else {
                constexpr int p = g - 16;
                o[0][p & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va[p].f, pf[p >> 2].f, o[0][p & 3], 0, 0, 0);
            }
            if constexpr (g + FG_VDIST >= 16 && g + FG_VDIST < 32) read_v(g + FG_VDIST - 16);
#if FG_PK
            {   // pairs whose first exponential belongs to the NEXT gap: [ (e1+1)/2, (e2+1)/2 )
                constexpr int e2 = exps_before(g + 2);
                constexpr int f0 = (e1 + 1) / 2, f1 = (e2 + 1) / 2;
                if constexpr (f1 > f0) do_fma(f0);
                if constexpr (f1 > f0 + 1) do_fma(f0 + 1);
            }
#endif
            // pack unit u in the first gap that starts with both of its exponentials issued (after this gap's MFMA: the
            // packed word is first used by a later gap's MFMA)
            constexpr int u0 = ep / 2, u1 = e0 / 2;
            if constexpr (u1 > u0) do_pack(u0);
            if constexpr (u1 > u0 + 1) do_pack(u0 + 1);
            if constexpr (e1 > e0) do_exp(e0);
            if constexpr (e1 > e0 + 1) do_exp(e0 + 1);
            if constexpr (e0 > ep) do_add(ep);
            if constexpr (e0 > ep + 1) do_add(ep + 1);
            if constexpr (!kTail && g >= 17) {      // 15 max3 steps in gaps 17..31 (scores of tile t+1 complete after gap 15)
                constexpr int i = g - 17;
                if constexpr (i == 0) mt = max3(sn[0][0][0], sn[0][1][0], sn[0][0][1]);
                else mt = max3(mt, sn[0][1][i], sn[0][0][i + 1]);
            }
            if constexpr (g == 28) write_k(cur);
            if constexpr (g == 30) write_v(cur ^ 1);
            __builtin_amdgcn_sched_barrier(0);
        });
#if FG_PK
        l_run[0] += ps2.x + ps2.y;
#else
#if FG_DOT2
        ps1 = dot2c_bf16(pf[3].w[3], 0x3f803f80u, ps1);      // pair 15
#endif
        l_run[0] += ps0 + ps1;
#endif
        if (!kTail) {
            absorb_max(pair_max(fmaxf(mt, sn[0][1][15])));
        } else if (t + 1 < t_end) {
            if ((int64_t)(t + 2) * kBN > Nkv) mask_tile(sn, t + 1);      // ragged last tile (wave-uniform)
            absorb_max(tile_max(sn));
        }
        __syncthreads();
    };

    // ---- prologue: K(0), V(0), K(1) resident; scores(0) computed and their max taken
    f32x16 sA[QB][2], sB[QB][2];
    load_k(t_begin);
    load_v(t_begin);
    write_k(0);
    write_v(0);
    load_k(t_begin + 1);
    write_k(1);
    __syncthreads();
    qk_tile(sA, 0);
    if ((int64_t)(t_begin + 1) * kBN > Nkv) mask_tile(sA, t_begin);
    m_run[0] = tile_max(sA);
    __syncthreads();      // every wave is done with the first K tile before step 0 overwrites it

#ifdef FG_YOUNG_PRIO
    // static priority for the second-dispatched half of the workgroup (the arbitration loser at equal priority)
    if (wave >= WAVES / 2) __builtin_amdgcn_s_setprio(FG_YOUNG_PRIO);
#endif
    // fast steps: the next tile exists in this range and is full
    const int t_fast = min(t_end - 1, (int)(Nkv / kBN) - 1);
    int t = t_begin;
    for (; t + 1 < t_fast; t += 2) {
        step_sched(sA, sB, t, std::false_type{}, std::integral_constant<int, 0>{});
        step_sched(sB, sA, t + 1, std::false_type{}, std::integral_constant<int, 1>{});
    }
    for (; t < t_end; ++t) {      // <= 3 steps: odd fast step, ragged-next step, last step
        step_sched(sA, sB, t, std::true_type{}, std::integral_constant<int, -1>{});
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) sA[0][sub] = sB[0][sub];
    }
#else
    // One pipeline step for tile t: sc = scores(t) (ready), sn <- scores(t+1) while softmax(sc) runs, then PV(t).
    auto step = [&](f32x16 (&sc)[QB][2], f32x16 (&sn)[QB][2], int t) {
        const int cur = (t - t_begin) & 1;
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
        if ((int64_t)(t + 2) * kBN > Nkv && t + 1 < t_end) mask_tile(sn, t + 1);   // ragged last tile (wave-uniform)
#if FG_EXP != 1
        __syncthreads();
#endif
    };

    // ---- prologue: K(0), V(0), K(1) resident; scores(0) computed
    f32x16 sA[QB][2], sB[QB][2];
    load_k(t_begin);
    load_v(t_begin);
    write_k(0);
    write_v(0);
    load_k(t_begin + 1);
    write_k(1);
    __syncthreads();
    qk_tile(sA, 0);
    if ((int64_t)(t_begin + 1) * kBN > Nkv) mask_tile(sA, t_begin);
    __syncthreads();      // every wave is done with the first K tile before step 0 overwrites it

    int t = t_begin;
    for (; t + 1 < t_end; t += 2) {
        step(sA, sB, t);
        step(sB, sA, t + 1);
    }
    if (t < t_end) step(sA, sB, t);

#endif

    // ---- epilogue: lane holds O[query r][d = 32db + 8g + 4hh + 0..3]
#pragma unroll
    for (int qi = 0; qi < QB; ++qi) {
        const float l_tot = pair_sum(l_run[qi]);
        if (piece < 0) {      // direct: normalise and store bf16
            const float inv = 1.0f / l_tot;
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
        } else {              // piece: un-normalised partial + (max, sum) to the workspace
            const int row = (wave * QB + qi) * 32 + r;
            float* orow = P.o_part + ((int64_t)piece * kBM + row) * kD;
#pragma unroll
            for (int db = 0; db < 4; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 w4;
#pragma unroll
                    for (int j = 0; j < 4; ++j) w4[j] = o[qi][db][4 * g + j];
                    *reinterpret_cast<f32x4*>(orow + 32 * db + 8 * g + 4 * hh) = w4;
                }
            if (hh == 0) {
                float* ml = P.ml_part + ((int64_t)piece * kBM + row) * 2;
                ml[0] = m_run[qi];
                ml[1] = l_tot;
            }
        }
    }
}

// Merge the S partials of every split q-block: out = sum_s w_s O_s / sum_s w_s l_s, w_s = 2^((m_s - max m) * scale*log2e).
__global__ __launch_bounds__(256) void attn_combine_kernel(const AttnParams P) {
    // one workgroup per 8 rows of a split q-block: 32 lanes x float4 cover d = 128
    const int rows_per_blk = 8, groups = kBM / rows_per_blk;
    const int blk = blockIdx.x / groups, rowgrp = blockIdx.x % groups;      // blk = (bh, rq)
    const int rq = blk % P.R, bh = blk / P.R;
    const int b = bh / P.H, h = bh % P.H;
    const int qb = P.nfull + rq;
    const int lane32 = threadIdx.x & 31, row = rowgrp * rows_per_blk + (threadIdx.x >> 5);
    const int64_t piece0 = ((int64_t)bh * P.R + rq) * P.S;
    const int64_t qrow = (int64_t)qb * kBM + row;
    if (qrow >= P.Nq) return;
    float mmax = -INFINITY;
    for (int sp = 0; sp < P.S; ++sp) mmax = fmaxf(mmax, P.ml_part[((piece0 + sp) * kBM + row) * 2]);
    float denom = 0.f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int sp = 0; sp < P.S; ++sp) {
        const float* ml = P.ml_part + ((piece0 + sp) * kBM + row) * 2;
        const float w = fast_exp2((ml[0] - mmax) * P.scale_log2e);
        denom += w * ml[1];
        const f32x4 ov = *reinterpret_cast<const f32x4*>(P.o_part + ((piece0 + sp) * kBM + row) * kD + lane32 * 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += w * ov[j];
    }
    const float inv = 1.0f / denom;
    bf16x4 w4;
#pragma unroll
    for (int j = 0; j < 4; ++j) w4[j] = (bf16)(acc[j] * inv);
    *reinterpret_cast<bf16x4*>(P.out + ((int64_t)b * P.Nq + qrow) * ((int64_t)P.H * kD) + (int64_t)h * kD + lane32 * 4) = w4;
}

// ---- host-side choice of (R, S): greedy list-scheduling estimate of the makespan on `cus` compute units
struct SplitChoice { int R, S; int64_t ws_bytes; double makespan; };

static double simulate_makespan(int64_t n_direct, int64_t n_pieces, double piece_cost, int cus) {
    // Greedy list scheduling in dispatch order: unit-cost direct workgroups first, then the pieces.
    const int64_t full = n_direct / cus, rem = n_direct % cus;
    if (n_pieces == 0) return (double)(full + (rem ? 1 : 0));
    std::priority_queue<double, std::vector<double>, std::greater<double>> free_at;
    for (int i = 0; i < cus; ++i) free_at.push((double)(full + (i < rem ? 1 : 0)));
    double last = (double)(full + (rem ? 1 : 0));
    for (int64_t i = 0; i < n_pieces; ++i) {
        const double t = free_at.top() + piece_cost;
        free_at.pop();
        free_at.push(t);
        last = t > last ? t : last;
    }
    return last;
}

static SplitChoice choose_split_uncached(int B, int64_t Nq, int64_t Nkv, int H, int cus, int64_t ws_limit) {
    const int64_t nqb = (Nq + kBM - 1) / kBM, bh = (int64_t)B * H;
    const int nt = (int)((Nkv + kBN - 1) / kBN);
    const int64_t per_piece = (int64_t)kBM * kD * 4 + (int64_t)kBM * 2 * 4;
    SplitChoice best{0, 1, 0, simulate_makespan(bh * nqb, 0, 0.0, cus)};
    const double base = best.makespan;
    const int s_list[] = {2, 3, 4, 5, 6, 8, 12, 16};
    for (int mode = 0; mode < 2; ++mode) {      // 0: split only the last q-block of every head; 1: split every q-block
        const int64_t R = mode == 0 ? 1 : nqb;
        if (mode == 0 && nqb < 2) continue;
        if (mode == 1 && bh * nqb > 4 * (int64_t)cus) continue;      // many blocks: only the tail needs balancing
        for (int S : s_list) {
            if (S * 8 > nt) break;      // at least 8 KV tiles per piece, or the prologue / merge dominates
            const int64_t pieces = bh * R * S;
            const int64_t ws = pieces * per_piece;
            if (ws > ws_limit) continue;
            // a piece costs its share of the KV range plus prologue/epilogue (~3 KV tiles) plus its share of the merge
            const double cost = 1.0 / S + 3.0 / nt + 0.004;
            const double ms = simulate_makespan(bh * (nqb - R), pieces, cost, cus);
            if (ms < best.makespan) best = SplitChoice{(int)R, S, ws, ms};
        }
    }
    if (best.makespan > 0.98 * base) best = SplitChoice{0, 1, 0, base};      // not worth a second launch
    return best;
}

// memo of a pure function (shapes repeat thousands of times per clip)
static SplitChoice choose_split(int B, int64_t Nq, int64_t Nkv, int H, int cus, int64_t ws_limit) {
    struct Key { int B, H, cus; int64_t Nq, Nkv, ws; SplitChoice val; };
    static thread_local Key memo[8];
    static thread_local int next = 0, filled = 0;
    for (int i = 0; i < filled; ++i) {
        const Key& e = memo[i];
        if (e.B == B && e.H == H && e.cus == cus && e.Nq == Nq && e.Nkv == Nkv && e.ws == ws_limit) return e.val;
    }
    const SplitChoice val = choose_split_uncached(B, Nq, Nkv, H, cus, ws_limit);
    memo[next] = Key{B, H, cus, Nq, Nkv, ws_limit, val};
    next = (next + 1) % 8;
    if (filled < 8) ++filled;
    return val;
}

static int device_cus() {
    static thread_local int cached_dev = -1, cached_cus = 256;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (dev != cached_dev) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cached_cus = n;
        cached_dev = dev;
    }
    return cached_cus * (256 / kBM);      // concurrent workgroup slots (two 128-row workgroups share a CU)
}

}  // namespace

extern "C" int64_t fg_attn_workspace_bytes(int B, int64_t Nq, int64_t Nkv, int H) {
    if (B <= 0 || Nq <= 0 || Nkv <= 0 || H <= 0) return 0;
    return choose_split(B, Nq, Nkv, H, device_cus(), (int64_t)1 << 40).ws_bytes;
}

extern "C" int fg_attn_split_choice(int B, int64_t Nq, int64_t Nkv, int H, int64_t workspace_bytes, int* R, int* S) {
    FG_CHECK_ARG(B > 0 && Nq > 0 && Nkv > 0 && H > 0 && R && S, "fg_attn_split_choice: bad arguments");
    const SplitChoice sc = choose_split(B, Nq, Nkv, H, device_cus(), workspace_bytes);
    *R = sc.R;
    *S = sc.S;
    return FG_OK;
}

extern "C" int fg_attn_fwd_bf16(const void* q, int64_t ldq, const void* k, int64_t ldk, const void* v, int64_t ldv,
                                void* out, int B, int64_t Nq, int64_t Nkv, int H, int D, float scale, void* workspace,
                                int64_t workspace_bytes, fg_stream_t stream) {
    FG_CHECK_ARG(q && k && v && out, "fg_attn_fwd_bf16: null pointer");
    FG_CHECK_ARG(D == kD, "fg_attn_fwd_bf16: only head_dim 128 is supported (got %d)", D);
    FG_CHECK_ARG(B > 0 && H > 0 && Nq > 0 && Nkv > 0, "fg_attn_fwd_bf16: B, H, Nq, Nkv must be positive");
    FG_CHECK_ARG(scale > 0.f, "fg_attn_fwd_bf16: scale must be positive");
    const int64_t hd = (int64_t)H * D;
    FG_CHECK_ARG(ldq >= hd && ldk >= hd && ldv >= hd && ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0,
                 "fg_attn_fwd_bf16: leading dimensions must be >= H*D and multiples of 8");
    FG_CHECK_ARG(FG_ALIGNED16(q) && FG_ALIGNED16(k) && FG_ALIGNED16(v) && FG_ALIGNED16(out) && FG_ALIGNED16(workspace),
                 "fg_attn_fwd_bf16: pointers must be 16-byte aligned");
    FG_CHECK_ARG(workspace_bytes >= 0 && (workspace != nullptr || workspace_bytes == 0), "fg_attn_fwd_bf16: bad workspace");
    const int64_t nqb = (Nq + kBM - 1) / kBM;
    FG_CHECK_ARG((Nkv + 2 * kBN) * ldk * 2 < (1ll << 32) && (Nkv + 2 * kBN) * ldv * 2 < (1ll << 32),
                 "fg_attn_fwd_bf16: K/V of one batch element must span < 4 GiB (32-bit buffer offsets)");
    const SplitChoice sc = choose_split(B, Nq, Nkv, H, device_cus(), workspace_bytes);
    AttnParams P;
    P.q = (const bf16*)q; P.k = (const bf16*)k; P.v = (const bf16*)v; P.out = (bf16*)out;
    P.ldq = ldq; P.ldk = ldk; P.ldv = ldv; P.Nq = Nq; P.Nkv = Nkv; P.H = H;
    P.nqb = (int)nqb; P.R = sc.R; P.S = sc.S; P.nfull = (int)(nqb - sc.R);
    const int64_t total_direct = (int64_t)B * H * P.nfull, pieces = (int64_t)B * H * sc.R * sc.S;
    FG_CHECK_ARG(total_direct + pieces < (1ll << 30), "fg_attn_fwd_bf16: grid too large");
    P.total_direct = (int)total_direct;
    P.scale_log2e = scale * 1.4426950408889634f;
    P.o_part = (float*)workspace;
    P.ml_part = (float*)workspace + pieces * kBM * kD;
#if FG_ATTN_VARIANT == 4
    hipLaunchKernelGGL((attn_fwd_kernel<4, 1>), dim3((unsigned)(total_direct + pieces)), dim3(256), 0, (hipStream_t)stream, P);
#else
    if (Nkv <= 1024)
        hipLaunchKernelGGL((attn_fwd_kernel<8, 1, true>), dim3((unsigned)(total_direct + pieces)), dim3(512), 0, (hipStream_t)stream, P);
    else
        hipLaunchKernelGGL((attn_fwd_kernel<8, 1, false>), dim3((unsigned)(total_direct + pieces)), dim3(512), 0, (hipStream_t)stream, P);
#endif
    if (int e = fg_launch_status("fg_attn_fwd_bf16")) return e;
    if (pieces > 0) {
        hipLaunchKernelGGL(attn_combine_kernel, dim3((unsigned)((int64_t)B * H * sc.R * (kBM / 8))), dim3(256), 0, (hipStream_t)stream, P);
        return fg_launch_status("fg_attn_fwd_bf16 (combine)");
    }
    return FG_OK;
}

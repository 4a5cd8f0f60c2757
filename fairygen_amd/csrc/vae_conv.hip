// Causal 3-D convolution of the Wan2.2 VAE decoder as an implicit GEMM on bf16 MFMA (gfx950).
//
// Replaces CausalConv3d.forward (models/wan_video_vae.py:33-52) incl. its feature-cache concat (the input
// tensor simply carries its kt-1 history frames in front), and the
// nearest-exact-2x Upsample + Conv2d pair of Resample38 (:242-251), the channel->time interleave after
// time_conv (:153-156) and the residual add of ResidualBlock (:301).
//
// Layout: activations channels-last (T,H,W,C) so that the reduction (tap, cin) is contiguous over cin;
// weights pre-packed once to [tap][Cout_pad][Cin_pad] (cin contiguous) by fg_conv_pack_weight_bf16.
// GEMM view: D[cout][pixel] = sum_{tap,cin} Wp[tap][cout][cin] * X[pixel shifted by tap][cin]
//   (weights are the MFMA A operand, pixels the B operand: the accumulator then holds, per lane, ONE pixel
//    and 4 consecutive couts per register group -> 8-byte channels-last stores, lane-local bias/residual).
// Tile: 128 couts x 128 pixels x 64 cin per step, 4 waves (2x2), each 64x64 = 2x2 v_mfma_f32_32x32x16_bf16
// accumulators; A/B tiles staged global->reg->LDS, double buffered, one barrier per step; LDS rows are 128 B
// with the 16-B chunk index XOR-swizzled by (row>>1)&7 -> conflict-free ds_read_b128.
// Roofline: MFMA (≈4000 FLOP/B at 704x1280).
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int kBMc = 128;   // couts per tile
constexpr int kBNp = 128;   // pixels per tile
constexpr int kBK = 64;     // cin per step
constexpr int kTile = 128 * kBK * 2;   // 16 KiB per operand tile

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

struct ConvParams {
    const bf16* x;          // (T + kt-1, Hin, Win, Cin): the first kt-1 frames are the causal history (feature cache)
    const bf16* wp;
    const bf16* bias;
    const bf16* residual;
    bf16* out;
    int T, H, W, Hin, Win, Cin, Cout, kt, ks, cin_pad, cout_pad, upsample, downsample, interleave;
    int64_t M;              // T*H*W output pixels
    uint32_t x_bytes, w_bytes;
};

__global__ __launch_bounds__(256, 2) void conv3d_cl_kernel(ConvParams p) {
    __shared__ __attribute__((aligned(16))) char smem[4 * kTile];   // A0 A1 B0 B1
    char* const a_lds = smem;               // weights tile
    char* const b_lds = smem + 2 * kTile;   // pixel tile

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

    const int n_ctiles = p.cout_pad / kBMc;
    const int ctile = blockIdx.x % n_ctiles;
    const int64_t ptile = blockIdx.x / n_ctiles;
    const int c0 = ctile * kBMc;
    const int64_t m0 = ptile * kBNp;

    // ---- staging roles: thread moves rows (tid>>3)+32*i, 16-byte chunk (tid&7) of both tiles.
    // Loads are bounds-checked buffer loads with 32-bit byte offsets: an out-of-range offset (spatial zero padding,
    // pixels past M, channels past Cin) returns zeros; the weight side needs NO per-step vector arithmetic at all
    // (loop-invariant voffset + scalar soffset).
    const int st_row = tid >> 3, st_chunk = tid & 7;
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.wp), 0, p.w_bytes, 0x00020000);
    constexpr uint32_t kOOB = 0xF0000000u;      // >= x_bytes (host-checked); + the small scalar offset cannot wrap
    int pt[4], py[4], px[4];
    bool pv[4];
    uint32_t w_off[4];
    int st_lds[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t m = m0 + st_row + 32 * i;
        pv[i] = m < p.M;
        const int64_t mm = m < p.M ? m : 0;
        px[i] = (int)(mm % p.W);
        py[i] = (int)((mm / p.W) % p.H);
        pt[i] = (int)(mm / ((int64_t)p.W * p.H));
        w_off[i] = (uint32_t)(((c0 + st_row + 32 * i) * p.cin_pad + st_chunk * 8) * 2);
        st_lds[i] = lds_off(st_row + 32 * i, st_chunk);
    }
    // stride-2 "downsample" mode = ZeroPad2d((0,1,0,1)) + Conv2d(stride 2): input row 2y + dy, no leading pad
    const int pad = p.downsample ? 0 : (p.ks >> 1);
    const int sstride = p.downsample ? 2 : 1;
    const int ksteps_per_tap = p.cin_pad / kBK;
    const int ntaps = p.kt * p.ks * p.ks;
    const int nsteps = ntaps * ksteps_per_tap;
    const uint32_t w_tap_stride = (uint32_t)(p.cout_pad * p.cin_pad * 2);

    // load cursor (runs one step ahead of the math): tap coordinates advance by counters, no divisions in the loop
    int ld_dt = 0, ld_dy = 0, ld_dx = 0, ld_cs = 0;
    uint32_t ld_wbase = 0;            // scalar byte offset of (tap, cin step) in the packed weights
    uint32_t x_off[4];
    auto tap_offsets = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int yy = py[i] * sstride + ld_dy - pad, xx = px[i] * sstride + ld_dx - pad;
            const bool ok = pv[i] && yy >= 0 && yy < p.H * sstride && xx >= 0 && xx < p.W * sstride;
            if (p.upsample) { yy >>= 1; xx >>= 1; }
            const uint32_t off = (uint32_t)((((pt[i] + ld_dt) * p.Hin + yy) * p.Win + xx) * p.Cin + st_chunk * 8) * 2u;
            x_off[i] = ok ? off : kOOB;
        }
    };
    u32x4 areg[4], breg[4];
    auto stage_load = [&]() {
        const uint32_t cs_bytes = (uint32_t)ld_cs * (kBK * 2);
        const bool cin_ok = ld_cs * kBK + st_chunk * 8 < p.Cin;      // dead chunks only when Cin % 64 != 0 (zero weights there)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            areg[i] = __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_off[i], ld_wbase + cs_bytes, 0);
            breg[i] = __builtin_amdgcn_raw_buffer_load_b128(x_rsrc, cin_ok ? x_off[i] : kOOB, cs_bytes, 0);
        }
    };
    auto advance = [&]() {
        if (++ld_cs == ksteps_per_tap) {
            ld_cs = 0;
            ld_wbase += w_tap_stride;
            if (++ld_dx == p.ks) {
                ld_dx = 0;
                if (++ld_dy == p.ks) { ld_dy = 0; ++ld_dt; }
            }
            tap_offsets();
        }
    };
    auto stage_write = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<u32x4*>(a_lds + buf * kTile + st_lds[i]) = areg[i];
            *reinterpret_cast<u32x4*>(b_lds + buf * kTile + st_lds[i]) = breg[i];
        }
    };

    // loop-invariant fragment read addresses
    int a_rd[2][4], b_rd[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            a_rd[i][ks] = lds_off(wm * 64 + i * 32 + r, 2 * ks + hh);
            b_rd[i][ks] = lds_off(wn * 64 + i * 32 + r, 2 * ks + hh);
        }

    f32x16 acc[2][2];   // [cout sub-tile mi][pixel sub-tile ni]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[a][b][j] = 0.f;

    tap_offsets();
    stage_load();
    stage_write(0);
    __syncthreads();

    for (int step = 0; step < nsteps; ++step) {
        const int cur = step & 1;
        const bool has_next = step + 1 < nsteps;      // wave-uniform
        if (has_next) {
            advance();
            stage_load();
        }
        const char* ab = a_lds + cur * kTile;
        const char* bb = b_lds + cur * kTile;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8 af[2], bfg[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i] = *reinterpret_cast<const bf16x8*>(ab + a_rd[i][ks]);
                bfg[i] = *reinterpret_cast<const bf16x8*>(bb + b_rd[i][ks]);
            }
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi], bfg[ni], acc[mi][ni], 0, 0, 0);
        }
        if (has_next) stage_write(cur ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane owns pixel column r of each pixel sub-tile; couts 8g + 4hh + 0..3 per register group
    const int cout2 = p.interleave ? p.Cout / 2 : p.Cout;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int64_t m = m0 + wn * 64 + ni * 32 + r;
        if (m >= p.M) continue;
        int64_t opix = m;   // output pixel index in the (possibly time-interleaved) output
        int64_t frame_stride = 0;
        if (p.interleave) {
            const int64_t hw = (int64_t)p.H * p.W;
            const int64_t t = m / hw;
            opix = (2 * t) * hw + (m - t * hw);
            frame_stride = hw;
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = c0 + wm * 64 + mi * 32 + 8 * g + 4 * hh;
                if (co >= p.Cout) continue;
                int oc = co;
                int64_t op = opix;
                if (p.interleave && co >= cout2) { oc = co - cout2; op = opix + frame_stride; }
                const bf16x4 bv = *reinterpret_cast<const bf16x4*>(p.bias + co);
                float o4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) o4[j] = rbf(acc[mi][ni][4 * g + j] + (float)bv[j]);
                bf16* dst = p.out + op * cout2 + oc;
                if (p.residual != nullptr) {
                    const bf16x4 rv = *reinterpret_cast<const bf16x4*>(p.residual + op * cout2 + oc);
#pragma unroll
                    for (int j = 0; j < 4; ++j) o4[j] += (float)rv[j];
                }
                bf16x4 w4;
#pragma unroll
                for (int j = 0; j < 4; ++j) w4[j] = (bf16)o4[j];
                *reinterpret_cast<bf16x4*>(dst) = w4;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Large-layer variant: 256 couts x 256 pixels x 64 cin per step, 8 waves (2 x 4), 128 x 64 per wave (4 x 2 MFMA tiles).
// At 128x128 the tile needs 64 B/clk/CU of operand traffic at full MFMA rate -- the whole vector-memory path -- and the
// register-staged ds_write_b128 path (~79 B/clk/CU) is slower still; a 256x256 tile needs 32 B/clk and the operands go
// global -> LDS directly (LDS-DMA, `buffer_load_dwordx4 ... lds`), bypassing VGPRs and the ds_write path.  An LDS-DMA
// wave-instruction writes 1 KiB linearly (8 rows x 128 B), so the XOR swizzle is applied to the per-lane SOURCE chunk;
// out-of-range lanes (zero padding) write zeros (probed: tools/probes/lds_dma_oob.hip).  2 stages x 64 KiB of dynamic
// LDS, one `vmcnt(0)` + barrier per step (the simple glds structure of cdna_hip_programming.md §5).
constexpr int kT2 = 256;                       // tile rows (couts and pixels)
constexpr int kConv256Default = 25681;         // form of the 256x256 kernel launched by default (see fg_conv3d_cl_bf16)
constexpr int kTile2 = kT2 * kBK * 2;          // 32 KiB per operand tile
typedef __attribute__((address_space(3))) void lds_void_t;

// WAVES = 8: 2 x 4 waves of 128 couts x 64 pixels (two waves per SIMD).  WAVES = 4: 2 x 2 waves of 128 x 128, one wave per
// SIMD with the 512-entry register file (256 accumulator registers): every fragment read from LDS feeds 4 MFMAs instead
// of 2 or 4, and the fragments of the next k-step are read while the current one's MFMAs run (PIPE), across the barrier.
template <int WAVES, bool PIPE>
__device__ __forceinline__ void conv3d_cl_256_body(const ConvParams& p) {
    constexpr int NI = WAVES == 8 ? 2 : 4;          // 32-pixel sub-tiles per wave
    constexpr int NP = 32 / WAVES;                  // 1-KiB LDS-DMA pieces per wave per operand tile
    extern __shared__ __attribute__((aligned(16))) char smem2[];      // A0 A1 B0 B1, 32 KiB each
    char* const a_lds = smem2;
    char* const b_lds = smem2 + 2 * kTile2;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int wm = WAVES == 8 ? wave >> 2 : wave >> 1, wn = WAVES == 8 ? (wave & 3) : (wave & 1);

    const int n_ctiles = p.cout_pad / kT2;
    const int ctile = blockIdx.x % n_ctiles;
    const int64_t ptile = blockIdx.x / n_ctiles;
    const int c0 = ctile * kT2;
    const int64_t m0 = ptile * kT2;

    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.wp), 0, p.w_bytes, 0x00020000);
    constexpr uint32_t kOOB = 0xF0000000u;

    // ---- LDS-DMA roles: wave w, instruction i (0..3) fills tile rows 8*(4w+i) .. +7; lane -> (row_local = lane>>3,
    // destination chunk c' = lane&7), source chunk = c' ^ swizzle(row)
    int pt[NP], py[NP], px[NP], src_chunk[NP];
    bool pv[NP];
    uint32_t w_off[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int row = 8 * (NP * wave + i) + (lane >> 3);
        src_chunk[i] = (lane & 7) ^ ((row >> 1) & 7);
        const int64_t m = m0 + row;
        pv[i] = m < p.M;
        const int64_t mm = pv[i] ? m : 0;
        px[i] = (int)(mm % p.W);
        py[i] = (int)((mm / p.W) % p.H);
        pt[i] = (int)(mm / ((int64_t)p.W * p.H));
        w_off[i] = (uint32_t)(((c0 + row) * p.cin_pad + src_chunk[i] * 8) * 2);
    }
    const int pad = p.downsample ? 0 : (p.ks >> 1);
    const int sstride = p.downsample ? 2 : 1;
    const int ksteps_per_tap = p.cin_pad / kBK;
    const int ntaps = p.kt * p.ks * p.ks;
    const int nsteps = ntaps * ksteps_per_tap;
    const uint32_t w_tap_stride = (uint32_t)(p.cout_pad * p.cin_pad * 2);

    int ld_dt = 0, ld_dy = 0, ld_dx = 0, ld_cs = 0;
    uint32_t ld_wbase = 0;
    uint32_t x_off[NP];
    auto tap_offsets = [&]() {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            int yy = py[i] * sstride + ld_dy - pad, xx = px[i] * sstride + ld_dx - pad;
            const bool ok = pv[i] && yy >= 0 && yy < p.H * sstride && xx >= 0 && xx < p.W * sstride;
            if (p.upsample) { yy >>= 1; xx >>= 1; }
            const uint32_t off = (uint32_t)((((pt[i] + ld_dt) * p.Hin + yy) * p.Win + xx) * p.Cin + src_chunk[i] * 8) * 2u;
            x_off[i] = ok ? off : kOOB;
        }
    };
    auto issue_dma = [&](int buf) {
        const uint32_t cs_bytes = (uint32_t)ld_cs * (kBK * 2);
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int dst = buf * kTile2 + 8 * (NP * wave + i) * 128;      // wave-uniform LDS byte offset of the 1 KiB piece
            const bool cin_ok = ld_cs * kBK + src_chunk[i] * 8 < p.Cin;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(w_rsrc, (lds_void_t*)(a_lds + dst), 16, w_off[i], ld_wbase + cs_bytes, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (lds_void_t*)(b_lds + dst), 16, cin_ok ? x_off[i] : kOOB, cs_bytes, 0, 0);
        }
    };
    auto advance = [&]() {
        if (++ld_cs == ksteps_per_tap) {
            ld_cs = 0;
            ld_wbase += w_tap_stride;
            if (++ld_dx == p.ks) {
                ld_dx = 0;
                if (++ld_dy == p.ks) { ld_dy = 0; ++ld_dt; }
            }
            tap_offsets();
        }
    };

    // (row >> 1) & 7 of the swizzle depends on r only (sub-tile bases are multiples of 32 rows): one base per k-step
    int a_rd[4], b_rd[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        a_rd[ks] = lds_off(wm * 128 + r, 2 * ks + hh);
        b_rd[ks] = lds_off(wn * (32 * NI) + r, 2 * ks + hh);
    }

    f32x16 acc[4][NI];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < NI; ++b)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[a][b][j] = 0.f;

    tap_offsets();
    issue_dma(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    auto read_frags = [&](bf16x8 (&af)[4], bf16x8 (&bfg)[NI], int buf, int ks) {
        const char* ab = a_lds + buf * kTile2;
        const char* bb = b_lds + buf * kTile2;
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8*>(ab + a_rd[ks] + i * 4096);
#pragma unroll
        for (int i = 0; i < NI; ++i) bfg[i] = *reinterpret_cast<const bf16x8*>(bb + b_rd[ks] + i * 4096);
    };
    auto mfma_block = [&](const bf16x8 (&af)[4], const bf16x8 (&bfg)[NI]) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi], bfg[ni], acc[mi][ni], 0, 0, 0);
    };
    if constexpr (!PIPE) {
        for (int step = 0; step < nsteps; ++step) {
            const int cur = step & 1;
            if (step + 1 < nsteps) {      // wave-uniform
                advance();
                issue_dma(cur ^ 1);
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                bf16x8 af[4], bfg[NI];
                read_frags(af, bfg, cur, ks);
                mfma_block(af, bfg);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's LDS-DMA pieces of the next stage have landed
            __syncthreads();
        }
    } else {
        // software pipeline over k-steps: fragments of k-step s+1 are in flight while the MFMAs of k-step s run; the last
        // k-step of a stage is multiplied AFTER the barrier, beside the first fragment reads of the next stage
        bf16x8 afA[4], bfA[NI], afB[4], bfB[NI];
        read_frags(afA, bfA, 0, 0);
        for (int step = 0; step < nsteps; ++step) {
            const int cur = step & 1;
            if (step + 1 < nsteps) {
                advance();
                issue_dma(cur ^ 1);
            }
            read_frags(afB, bfB, cur, 1);
            mfma_block(afA, bfA);
            read_frags(afA, bfA, cur, 2);
            mfma_block(afB, bfB);
            read_frags(afB, bfB, cur, 3);
            mfma_block(afA, bfA);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (step + 1 < nsteps) read_frags(afA, bfA, cur ^ 1, 0);
            mfma_block(afB, bfB);
        }
    }

    // ---- epilogue (same fragment map as the 128x128 kernel)
    const int cout2 = p.interleave ? p.Cout / 2 : p.Cout;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int64_t m = m0 + wn * (32 * NI) + ni * 32 + r;
        if (m >= p.M) continue;
        int64_t opix = m, frame_stride = 0;
        if (p.interleave) {
            const int64_t hw = (int64_t)p.H * p.W;
            const int64_t t = m / hw;
            opix = (2 * t) * hw + (m - t * hw);
            frame_stride = hw;
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = c0 + wm * 128 + mi * 32 + 8 * g + 4 * hh;
                if (co >= p.Cout) continue;
                int oc = co;
                int64_t op = opix;
                if (p.interleave && co >= cout2) { oc = co - cout2; op = opix + frame_stride; }
                const bf16x4 bv = *reinterpret_cast<const bf16x4*>(p.bias + co);
                float o4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) o4[j] = rbf(acc[mi][ni][4 * g + j] + (float)bv[j]);
                if (p.residual != nullptr) {
                    const bf16x4 rv = *reinterpret_cast<const bf16x4*>(p.residual + op * cout2 + oc);
#pragma unroll
                    for (int j = 0; j < 4; ++j) o4[j] += (float)rv[j];
                }
                bf16x4 w4;
#pragma unroll
                for (int j = 0; j < 4; ++j) w4[j] = (bf16)o4[j];
                *reinterpret_cast<bf16x4*>(p.out + op * cout2 + oc) = w4;
            }
    }
}

__global__ __launch_bounds__(512, 2) void conv3d_cl_256_kernel(ConvParams p) { conv3d_cl_256_body<8, false>(p); }
__global__ __launch_bounds__(512, 2) void conv3d_cl_256p_kernel(ConvParams p) { conv3d_cl_256_body<8, true>(p); }
__global__ __launch_bounds__(256, 1) void conv3d_cl_256w4_kernel(ConvParams p) { conv3d_cl_256_body<4, false>(p); }
__global__ __launch_bounds__(256, 1) void conv3d_cl_256w4p_kernel(ConvParams p) { conv3d_cl_256_body<4, true>(p); }

__global__ void pack_weight_kernel(const bf16* __restrict__ w, bf16* __restrict__ packed, int Cout, int Cin, int ntaps,
                                   int cout_pad, int cin_pad) {
    const int64_t total = (int64_t)ntaps * cout_pad * cin_pad;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % cin_pad);
        const int co = (int)((i / cin_pad) % cout_pad);
        const int tap = (int)(i / ((int64_t)cin_pad * cout_pad));
        bf16 val = (bf16)0.f;
        if (ci < Cin && co < Cout) val = w[((int64_t)co * Cin + ci) * ntaps + tap];
        packed[i] = val;
    }
}

inline int roundup(int a, int b) { return (a + b - 1) / b * b; }

// FAIRYGEN_CONV_TILE=128|256 forces a tile variant (A/B measurements); unset/0 = automatic.
inline int conv_variant_override() {
    static const int v = [] { const char* e = getenv("FAIRYGEN_CONV_TILE"); return e ? atoi(e) : 0; }();
    return v;
}

}  // namespace

extern "C" {

int64_t fg_conv_packed_bytes(int Cout, int Cin, int kt, int kh, int kw) {
    if (Cout <= 0 || Cin <= 0 || kt <= 0 || kh <= 0 || kw <= 0) return 0;
    return (int64_t)kt * kh * kw * roundup(Cout, kBMc) * roundup(Cin, kBK) * 2;
}

int fg_conv_pack_weight_bf16(const void* w, void* packed, int Cout, int Cin, int kt, int kh, int kw, fg_stream_t stream) {
    FG_CHECK_ARG(w && packed && Cout > 0 && Cin > 0 && kt > 0 && kh > 0 && kw > 0, "fg_conv_pack_weight_bf16: bad arguments");
    const int ntaps = kt * kh * kw;
    hipLaunchKernelGGL(pack_weight_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, (const bf16*)w, (bf16*)packed, Cout,
                       Cin, ntaps, roundup(Cout, kBMc), roundup(Cin, kBK));
    return fg_launch_status("fg_conv_pack_weight_bf16");
}

int fg_conv_tile_choice(int T, int H, int W, int Cout) {
    // the 256x256 LDS-DMA kernel when it fills the chip (>= one workgroup per CU) and Cout is a multiple of 256; the
    // 128x128 kernel (two workgroups per CU) for the low-resolution layers and odd channel counts.
    if (T <= 0 || H <= 0 || W <= 0 || Cout <= 0 || Cout % kT2 != 0) return 128;
    const int64_t M = (int64_t)T * H * W;
    const int64_t blocks256 = ((M + kT2 - 1) / kT2) * (roundup(Cout, kBMc) / kT2);
    const int variant = conv_variant_override() ? conv_variant_override() : (blocks256 >= 224 ? 256 : 128);
    return variant >= 256 ? 256 : 128;
}

int fg_conv3d_cl_bf16(const void* x, const void* w_packed, const void* bias, const void* residual, void* out, int T, int H,
                      int W, int Cin, int Cout, int kt, int ks, int resample, int time_interleave, fg_stream_t stream) {
    const int upsample2x = resample == 1, downsample2x = resample == 2;
    FG_CHECK_ARG(resample >= 0 && resample <= 2, "fg_conv3d_cl_bf16: resample must be 0 (none), 1 (nearest 2x up) or 2 (stride-2 down)");
    FG_CHECK_ARG(!downsample2x || (kt == 1 && ks == 3), "fg_conv3d_cl_bf16: stride-2 mode is the 3x3 Conv2d of Resample38 downsample");
    FG_CHECK_ARG(x && w_packed && bias && out, "fg_conv3d_cl_bf16: null pointer");
    FG_CHECK_ARG(T > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "fg_conv3d_cl_bf16: sizes must be positive");
    FG_CHECK_ARG((kt == 1 || kt == 3) && (ks == 1 || ks == 3), "fg_conv3d_cl_bf16: kt and ks must be 1 or 3");
    FG_CHECK_ARG(Cin % 8 == 0 && Cout % 4 == 0, "fg_conv3d_cl_bf16: need Cin %% 8 == 0 and Cout %% 4 == 0 (Cin=%d Cout=%d)", Cin, Cout);
    FG_CHECK_ARG(!upsample2x || (H % 2 == 0 && W % 2 == 0 && kt == 1), "fg_conv3d_cl_bf16: upsample2x needs even H, W and kt==1");
    FG_CHECK_ARG(!time_interleave || (Cout % 8 == 0), "fg_conv3d_cl_bf16: time_interleave needs Cout %% 8 == 0");
    FG_CHECK_ARG(FG_ALIGNED16(x) && FG_ALIGNED16(w_packed) && ((uintptr_t)bias & 7) == 0 && ((uintptr_t)residual & 7) == 0 &&
                     ((uintptr_t)out & 7) == 0,
                 "fg_conv3d_cl_bf16: misaligned pointer");
    ConvParams p;
    p.x = (const bf16*)x; p.wp = (const bf16*)w_packed; p.bias = (const bf16*)bias;
    p.residual = (const bf16*)residual; p.out = (bf16*)out;
    p.T = T; p.H = H; p.W = W;
    p.Hin = upsample2x ? H / 2 : (downsample2x ? 2 * H : H); p.Win = upsample2x ? W / 2 : (downsample2x ? 2 * W : W);
    p.Cin = Cin; p.Cout = Cout; p.kt = kt; p.ks = ks;
    p.cin_pad = roundup(Cin, kBK); p.cout_pad = roundup(Cout, kBMc);
    p.upsample = upsample2x ? 1 : 0; p.downsample = downsample2x ? 1 : 0; p.interleave = time_interleave ? 1 : 0;
    p.M = (int64_t)T * H * W;
    const int64_t x_bytes = (int64_t)(T + kt - 1) * p.Hin * p.Win * Cin * 2;
    const int64_t w_bytes = fg_conv_packed_bytes(Cout, Cin, kt, ks, ks);
    FG_CHECK_ARG(x_bytes < 0xF0000000ll && w_bytes < 0xF0000000ll, "fg_conv3d_cl_bf16: input / weights must be < 3.75 GiB (32-bit offsets)");
    p.x_bytes = (uint32_t)x_bytes; p.w_bytes = (uint32_t)w_bytes;
    const int64_t blocks256 = ((p.M + kT2 - 1) / kT2) * (p.cout_pad / kT2);
    if (fg_conv_tile_choice(T, H, W, Cout) == 256) {
        FG_CHECK_ARG(blocks256 < (1ll << 31), "fg_conv3d_cl_bf16: grid too large");
        // FAIRYGEN_CONV_TILE: 256 = default form (8 waves, fragments of the next k-step read beside the MFMAs of the current one and across the
        // barrier: +7-10 % on the main shapes); 2568 / 25681 / 2564 / 25641 force <8 waves> / <8, pipelined> / <4 waves> / <4, pipelined>
        // (the compiler-scheduled 4-wave forms measure 8-14 % SLOWER than the 8-wave ones: tools/conv_ab.sh)
        const int form = conv_variant_override() > 256 ? conv_variant_override() : kConv256Default;
#define FG_LAUNCH_256(KERNEL, W)                                                                                                 \
    do {                                                                                                                         \
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(KERNEL),                                \
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, 4 * kTile2);              \
        FG_CHECK_ARG(attr == hipSuccess, "fg_conv3d_cl_bf16: cannot reserve 128 KiB of LDS: %s", hipGetErrorString(attr));       \
        hipLaunchKernelGGL(KERNEL, dim3((unsigned)blocks256), dim3(W * 64), 4 * kTile2, (hipStream_t)stream, p);                 \
    } while (0)
        if (form == 2564) FG_LAUNCH_256(conv3d_cl_256w4_kernel, 4);
        else if (form == 25641) FG_LAUNCH_256(conv3d_cl_256w4p_kernel, 4);
        else if (form == 25681) FG_LAUNCH_256(conv3d_cl_256p_kernel, 8);
        else FG_LAUNCH_256(conv3d_cl_256_kernel, 8);
#undef FG_LAUNCH_256
        return fg_launch_status("fg_conv3d_cl_bf16 (256x256)");
    }
    const int64_t blocks = ((p.M + kBNp - 1) / kBNp) * (p.cout_pad / kBMc);
    FG_CHECK_ARG(blocks < (1ll << 31), "fg_conv3d_cl_bf16: grid too large");
    hipLaunchKernelGGL(conv3d_cl_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
    return fg_launch_status("fg_conv3d_cl_bf16");
}

}  // extern "C"

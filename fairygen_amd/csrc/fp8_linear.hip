// Activation side of the reference's fp8 Linear (AutoWrappedLinear.fp8_linear, core/vram/layers.py:321-357): per-row dynamic
// scale, division, cast to OCP e4m3 (torch.float8_e4m3fn).  The GEMM itself stays on the library (torch._scaled_mm ->
// hipBLASLt fp8 MFMA, row-wise scale_a, unit scale_b, bf16 bias), exactly the call the reference makes.
// HBM-bound: one wave owns one row, keeps it in registers as bf16 (C <= 14336 -> <= 28 x 16-byte vectors per lane), one
// wave reduction for the row maximum, every input byte read once, one byte written per element.
#include "common.h"

namespace {

constexpr int kRowsPerBlock = 4;

// Two fp32 -> two e4m3 bytes in the low half of a dword (v_cvt_pk_fp8_f32: RNE, OCP e4m3fn on gfx950).
__device__ __forceinline__ uint32_t cvt2_fp8(float a, float b) {
    return (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false) & 0xffffu;
}

template <int MAXV>
__global__ __launch_bounds__(256) void fp8_quant_rows_kernel(const bf16* __restrict__ x, int64_t ldx, uint8_t* __restrict__ out,
                                                             float* __restrict__ scale, bf16* __restrict__ act_out,
                                                             int64_t rows, int C, int act, float fp8_max) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * kRowsPerBlock + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nvec = C >> 3;
    const bf16* xr = x + row * ldx;
    bf16x8 v[MAXV];
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int vi = lane + i * 64;
        if (vi < nvec) {
            v[i] = *reinterpret_cast<const bf16x8*>(xr + (int64_t)vi * 8);
            if (act == 1) {      // GELU(tanh) of ffn.0's output, rounded to bf16 like the nn.GELU module's result
#pragma unroll
                for (int j = 0; j < 8; ++j) v[i][j] = (bf16)gelu_tanh_f((float)v[i][j]);
                if (act_out != nullptr) *reinterpret_cast<bf16x8*>(act_out + row * (int64_t)C + (int64_t)vi * 8) = v[i];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) amax = fmaxf(amax, fabsf((float)v[i][j]));
        }
    }
    amax = wave_max(amax);
    // scale_a = clamp(x_max / fp8_max, min=1).float(): the quotient is a bf16 tensor op in the reference (layers.py:338)
    const float s = fmaxf(rbf(amax / fp8_max), 1.0f);
    const float denom = s + 1e-8f;                       // input / (scale_a + 1e-8) in fp32 (:340)
    if (lane == 0) scale[row] = s;
    uint8_t* orow = out + row * (int64_t)C;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int vi = lane + i * 64;
        if (vi < nvec) {
            u32x2 w;
            w[0] = cvt2_fp8((float)v[i][0] / denom, (float)v[i][1] / denom) | (cvt2_fp8((float)v[i][2] / denom, (float)v[i][3] / denom) << 16);
            w[1] = cvt2_fp8((float)v[i][4] / denom, (float)v[i][5] / denom) | (cvt2_fp8((float)v[i][6] / denom, (float)v[i][7] / denom) << 16);
            *reinterpret_cast<u32x2*>(orow + (int64_t)vi * 8) = w;
        }
    }
}

}  // namespace

extern "C" int fg_fp8_quant_rows_bf16(const void* x, int64_t ldx, void* out_fp8, float* scale, void* act_out, int64_t rows,
                                      int C, int act, float fp8_max, fg_stream_t stream) {
    FG_CHECK_ARG(x && out_fp8 && scale, "fg_fp8_quant_rows_bf16: null pointer");
    FG_CHECK_ARG(rows >= 0 && C > 0 && C % 8 == 0 && C <= 28 * 64 * 8, "fg_fp8_quant_rows_bf16: C must be a multiple of 8, <= 14336 (got %d)", C);
    FG_CHECK_ARG(ldx >= C && ldx % 8 == 0 && FG_ALIGNED16(x) && (((uintptr_t)out_fp8) & 7) == 0 && FG_ALIGNED16(act_out),
                 "fg_fp8_quant_rows_bf16: x / act_out must be 16-byte aligned with ldx a multiple of 8, out_fp8 8-byte aligned");
    FG_CHECK_ARG(act == 0 || act == 1, "fg_fp8_quant_rows_bf16: act must be 0 (none) or 1 (gelu_tanh)");
    FG_CHECK_ARG(act == 1 || act_out == nullptr, "fg_fp8_quant_rows_bf16: act_out only with an activation");
    FG_CHECK_ARG(fp8_max > 0.f, "fg_fp8_quant_rows_bf16: fp8_max must be positive");
    if (rows == 0) return FG_OK;
    const dim3 grid((unsigned)((rows + kRowsPerBlock - 1) / kRowsPerBlock));
    if (C <= 8 * 64 * 8)
        hipLaunchKernelGGL(fp8_quant_rows_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)x, ldx, (uint8_t*)out_fp8,
                           scale, (bf16*)act_out, rows, C, act, fp8_max);
    else
        hipLaunchKernelGGL(fp8_quant_rows_kernel<28>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)x, ldx, (uint8_t*)out_fp8,
                           scale, (bf16*)act_out, rows, C, act, fp8_max);
    return fg_launch_status("fg_fp8_quant_rows_bf16");
}

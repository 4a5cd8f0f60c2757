// umT5 text-encoder helpers (run twice per clip, outside the denoise loop): biased / masked row softmax of the
// T5 attention and the gated tanh-GELU of the feed-forward, both with the reference's bf16 rounding points
// (models/wan_video_text_encoder.py:19-22,72-87,109-111).  GEMMs stay on hipBLASLt; T5LayerNorm = fg_rmsnorm_rope_bf16.
#include "common.h"

namespace {

// probs[r][c] = softmax_c( bf16(scores[r][c] + (key_mask[c] ? bias[r][c] : bf16_min)) ) in fp32, rounded to bf16.
// One 256-thread workgroup per row.
__global__ __launch_bounds__(256) void softmax_bias_kernel(const bf16* __restrict__ scores, const bf16* __restrict__ bias,
                                                           const int* __restrict__ key_mask, bf16* __restrict__ probs,
                                                           int64_t cols) {
    __shared__ float red[8];
    const int64_t row = blockIdx.x;
    const bf16* s = scores + row * cols;
    const bf16* b = bias + row * cols;
    const float kMin = -3.3895313892515355e38f;      // torch.finfo(torch.bfloat16).min
    auto val = [&](int64_t c) {
        const float bb = (key_mask == nullptr || key_mask[c] != 0) ? (float)b[c] : kMin;
        return rbf((float)s[c] + bb);
    };
    float mx = -INFINITY;
    for (int64_t c = threadIdx.x; c < cols; c += 256) mx = fmaxf(mx, val(c));
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int64_t c = threadIdx.x; c < cols; c += 256) sum += expf(val(c) - mx);
    sum = wave_sum(sum);
    if ((threadIdx.x & 63) == 0) red[4 + (threadIdx.x >> 6)] = sum;
    __syncthreads();
    sum = red[4] + red[5] + red[6] + red[7];
    for (int64_t c = threadIdx.x; c < cols; c += 256) probs[row * cols + c] = (bf16)(expf(val(c) - mx) / sum);
}

// out = fc1 * GELU(gate), GELU written out as the reference does: 0.5*x*(1+tanh(sqrt(2/pi)*(x+0.044715*x^3))),
// every tensor op rounded to bf16.
__global__ __launch_bounds__(256) void gated_gelu_kernel(const bf16* __restrict__ fc1, const bf16* __restrict__ gate,
                                                         bf16* __restrict__ out, int64_t nvec) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(fc1 + i * 8);
        const bf16x8 g = *reinterpret_cast<const bf16x8*>(gate + i * 8);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x = (float)g[j];
            const float p3 = rbf(rbf(x * x) * x);      // torch.pow(bf16, 3.0) = x*x*x with a bf16 rounding per multiply
            const float inner = rbf(x + rbf(0.044715f * p3));
            const float th = rbf(tanhf(rbf(0.7978845608028654f * inner)));
            const float gelu = rbf(rbf(0.5f * x) * rbf(1.0f + th));
            o[j] = (bf16)((float)a[j] * gelu);
        }
        *reinterpret_cast<bf16x8*>(out + i * 8) = o;
    }
}

}  // namespace

extern "C" {

int fg_softmax_bias_bf16(const void* scores, const void* bias, const int* key_mask, void* probs, int64_t rows, int64_t cols,
                         fg_stream_t stream) {
    FG_CHECK_ARG(scores && bias && probs && rows > 0 && cols > 0 && rows < (1ll << 31), "fg_softmax_bias_bf16: bad arguments");
    hipLaunchKernelGGL(softmax_bias_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (const bf16*)scores,
                       (const bf16*)bias, key_mask, (bf16*)probs, cols);
    return fg_launch_status("fg_softmax_bias_bf16");
}

int fg_gated_gelu_bf16(const void* fc1, const void* gate, void* out, int64_t n, fg_stream_t stream) {
    FG_CHECK_ARG(fc1 && gate && out && n >= 0 && n % 8 == 0, "fg_gated_gelu_bf16: n must be a multiple of 8");
    FG_CHECK_ARG(FG_ALIGNED16(fc1) && FG_ALIGNED16(gate) && FG_ALIGNED16(out), "fg_gated_gelu_bf16: misaligned pointer");
    if (n == 0) return FG_OK;
    const int64_t nvec = n / 8;
    const unsigned grid = (unsigned)((nvec + 255) / 256 < 8192 ? (nvec + 255) / 256 : 8192);
    hipLaunchKernelGGL(gated_gelu_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const bf16*)fc1, (const bf16*)gate,
                       (bf16*)out, nvec);
    return fg_launch_status("fg_gated_gelu_bf16");
}

}  // extern "C"

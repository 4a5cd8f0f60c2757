// Shared device/host helpers for libfairygen_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/fairygen_hip.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define FG_WAVE 64

void fg_set_error(const char* fmt, ...);

#define FG_CHECK_ARG(cond, ...)                                  \
    do {                                                         \
        if (!(cond)) {                                           \
            fg_set_error(__VA_ARGS__);                           \
            return FG_EINVAL;                                    \
        }                                                        \
    } while (0)

#define FG_ALIGNED16(p) ((((uintptr_t)(p)) & 15) == 0)

static inline int fg_launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        fg_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return FG_ELAUNCH;
    }
    return FG_OK;
}

// Round-to-nearest-even through bf16 (mirrors one PyTorch bf16 op boundary).
__device__ __forceinline__ float rbf(float x) { return (float)(bf16)x; }

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float gelu_tanh_f(float x) {
    const float kBeta = 0.7978845608028654f, kKappa = 0.044715f;
    const float u = kBeta * (x + kKappa * x * x * x);
    // tanh(u) = 1 - 2/(1+exp(2u)); clamp keeps exp finite.
    const float e = __expf(2.0f * fminf(u, 15.0f));
    const float t = 1.0f - 2.0f / (1.0f + e);
    return 0.5f * x * (1.0f + t);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Block-wide sum for blockDim.x == NW*64; red must hold NW floats. Returns the total in every thread.
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    if (NW == 1) return v;
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) t += red[i];
    return t;
}

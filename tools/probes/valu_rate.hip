// Issue-rate probe for gfx950: cycles per wave64 instruction of v_fma_f32, v_exp_f32, v_pk_fma_f32, v_pk_add_f32, v_cvt_pk_bf16_f32,
// v_max3_f32 and the MFMA 32x32x16 bf16, for 1 and 2 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int kIters = 2000, kUnroll = 16;

template <int OP>
__global__ void probe(float* out, long long* cycles) {
    float a[kUnroll];
    f32x2 p[kUnroll];
    for (int i = 0; i < kUnroll; ++i) { a[i] = threadIdx.x * 0.001f + i; p[i] = f32x2{a[i], a[i] + 1.f}; }
    const float s = out[0], c = out[1];
    const f32x2 s2 = {s, s}, c2 = {c, c};
    f32x16 acc[4] = {};
    bf16x8 fa = {}, fb = {};
    __syncthreads();
    const long long t0 = clock64();
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int i = 0; i < kUnroll; ++i) {
            if constexpr (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(s), "v"(c));
            if constexpr (OP == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            if constexpr (OP == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(s2), "v"(c2));
            if constexpr (OP == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
            if constexpr (OP == 4) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if constexpr (OP == 5) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(s), "v"(c));
            if constexpr (OP == 6) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if constexpr (OP == 7) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[i & 3], 0, 0, 0);
            if constexpr (OP == 8) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(s2));
            if constexpr (OP == 9) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
        }
    }
    const long long t1 = clock64();
    float r = 0.f;
    for (int i = 0; i < kUnroll; ++i) r += a[i] + p[i].x + p[i].y;
    for (int i = 0; i < 4; ++i) r += acc[i][0];
    out[2 + blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int OP>
void run(const char* name, float* out, long long* cyc) {
    for (int threads : {64, 256, 512}) {
        hipLaunchKernelGGL(probe<OP>, dim3(1), dim3(threads), 0, 0, out, cyc);
        hipDeviceSynchronize();
        long long c;
        hipMemcpy(&c, cyc, sizeof(c), hipMemcpyDeviceToHost);
        printf("%-20s threads %3d (%d wave/SIMD): %.2f clock64 ticks per instruction per wave\n", name, threads,
               threads <= 256 ? 1 : 2, (double)c / ((double)kIters * kUnroll));
    }
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 1 << 16); hipMalloc(&cyc, 64);
    float init[2] = {1.0001f, 0.5f};
    hipMemcpy(out, init, sizeof(init), hipMemcpyHostToDevice);
    run<0>("v_fma_f32", out, cyc);
    run<6>("v_add_f32", out, cyc);
    run<1>("v_exp_f32", out, cyc);
    run<9>("v_rcp_f32", out, cyc);
    run<2>("v_pk_fma_f32", out, cyc);
    run<3>("v_pk_add_f32", out, cyc);
    run<8>("v_pk_mul_f32", out, cyc);
    run<4>("v_cvt_pk_bf16_f32", out, cyc);
    run<5>("v_max3_f32", out, cyc);
    run<7>("mfma_32x32x16_bf16", out, cyc);
    return 0;
}

// Probe: operand lane map and issue rate of v_mfma_f32_32x32x64_f8f6f4 (e4m3 x e4m3) on gfx950.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_f8f6f4.hip -o tools/probes/mfma_f8f6f4 && tools/probes/mfma_f8f6f4
// Hypothesis checked with exact integer data: lane l (r = l & 31, h = l >> 5) holds A[row r][k = 32h + j] and B[k = 32h + j][col r],
// j = 0..31, byte j of its 8 operand registers; C/D as the bf16 32x32 forms.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ void one_mfma(const uint8_t* a, const uint8_t* b, float* d, int scaled) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    v8i av, bv;
    for (int i = 0; i < 8; ++i) {
        av[i] = *reinterpret_cast<const int*>(a + r * 64 + 32 * h + 4 * i);      // A row-major [32][64]
        bv[i] = *reinterpret_cast<const int*>(b + r * 64 + 32 * h + 4 * i);      // B^T row-major [32 cols][64 k]
    }
    v16f c = {};
    if (scaled) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    else asm volatile("v_mfma_f32_32x32x64_f8f6f4 %0, %1, %2, 0\n\ts_nop 15\n\ts_nop 7" : "=v"(c) : "v"(av), "v"(bv));
    for (int j = 0; j < 16; ++j) d[((j & 3) + 8 * (j >> 2) + 4 * h) * 32 + r] = c[j];      // row = (reg&3)+8(reg>>2)+4h, col = r
}

template <int KIND>
__global__ __launch_bounds__(256) void rate(float* out, int iters) {
    v8i av, bv;
    for (int i = 0; i < 8; ++i) { av[i] = 0x38383838 + threadIdx.x; bv[i] = 0x38383838; }
    v16f c0 = {}, c1 = {}, c2 = {}, c3 = {};
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {
            asm volatile("v_mfma_f32_32x32x64_f8f6f4 %0, %4, %5, %0\n\tv_mfma_f32_32x32x64_f8f6f4 %1, %4, %5, %1\n\t"
                         "v_mfma_f32_32x32x64_f8f6f4 %2, %4, %5, %2\n\tv_mfma_f32_32x32x64_f8f6f4 %3, %4, %5, %3"
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(av), "v"(bv));
        } else if (KIND == 1) {
            c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c0, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c1, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c2, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c3, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        } else {
            typedef short v8s __attribute__((ext_vector_type(8)));
            v8s a2, b2;
            for (int i = 0; i < 8; ++i) { a2[i] = (short)av[i]; b2[i] = (short)bv[i]; }
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %4, %5, %1\n\t"
                         "v_mfma_f32_32x32x16_bf16 %2, %4, %5, %2\n\tv_mfma_f32_32x32x16_bf16 %3, %4, %5, %3"
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a2), "v"(b2));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

static uint8_t e4m3(int v) {      // small integers -8..8 -> OCP e4m3fn (bias 7), exact
    if (v == 0) return 0;
    const uint8_t s = v < 0 ? 0x80 : 0;
    int m = abs(v), e = 0;
    while ((1 << (e + 1)) <= m) ++e;                 // m = 2^e * (1 + f/8)
    const int frac = ((m << 3) >> e) & 7;
    return s | (uint8_t)((e + 7) << 3) | (uint8_t)frac;
}

int main() {
    uint8_t ha[32 * 64], hb[32 * 64];
    int ia[32 * 64], ib[32 * 64];
    srand(1);
    for (int i = 0; i < 32 * 64; ++i) { ia[i] = rand() % 9 - 4; ib[i] = rand() % 7 - 3; ha[i] = e4m3(ia[i]); hb[i] = e4m3(ib[i]); }
    uint8_t *da, *db; float* dd;
    hipMalloc(&da, sizeof ha); hipMalloc(&db, sizeof hb); hipMalloc(&dd, 32 * 32 * 4);
    hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
    for (int scaled = 0; scaled < 2; ++scaled) {
        hipMemset(dd, 0, 32 * 32 * 4);
        one_mfma<<<1, 64>>>(da, db, dd, scaled);
        float hd[32 * 32];
        hipMemcpy(hd, dd, sizeof hd, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            int s = 0;
            for (int k = 0; k < 64; ++k) s += ia[i * 64 + k] * ib[j * 64 + k];
            if (hd[i * 32 + j] != (float)s) { if (bad < 4) printf("  D[%d][%d] = %g, want %d\n", i, j, hd[i * 32 + j], s); ++bad; }
        }
        printf("%s v_mfma 32x32x64 f8f6f4 (e4m3): %d of 1024 outputs wrong with the contiguous-32-k lane map\n", scaled ? "scaled" : "plain", bad);
    }
    float* out; hipMalloc(&out, 1024 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int kind = 0; kind < 3; ++kind) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (kind == 0) rate<0><<<256, 256>>>(out, iters); else if (kind == 1) rate<1><<<256, 256>>>(out, iters); else rate<2><<<256, 256>>>(out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flop = 256.0 * 4 * iters * 4 * 2.0 * 32 * 32 * (kind == 2 ? 16 : 64);
            if (rep) printf("%s: %.3f ms, %.0f TFLOP/s (1 workgroup of 4 waves per CU, 4 independent accumulators)\n",
                            kind == 0 ? "plain f8f6f4 32x32x64" : kind == 1 ? "scaled f8f6f4 32x32x64" : "bf16 32x32x16", ms, flop / ms / 1e9);
        }
    }
    return 0;
}

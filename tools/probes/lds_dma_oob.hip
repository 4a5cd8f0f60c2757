// Probe: does an out-of-range `buffer_load_dwordx4 ... lds` (LDS-DMA) write zeros into LDS, or leave it untouched?
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void lds_void;
__global__ void probe(const float* g, float* o, int n_bytes) {
    __shared__ __attribute__((aligned(16))) float lds[256];
    lds[threadIdx.x] = -7.0f; lds[threadIdx.x + 64] = -7.0f; lds[threadIdx.x + 128] = -7.0f; lds[threadIdx.x + 192] = -7.0f;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g), 0, n_bytes, 0x00020000);
    // lanes 0..31 in range, lanes 32..63 out of range
    const uint32_t off = threadIdx.x < 32 ? threadIdx.x * 16 : 0xF0000000u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_void*)lds, 16, off, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = 0; i < 4; ++i) o[threadIdx.x * 4 + i] = lds[threadIdx.x * 4 + i];
}
int main() {
    float *g, *o, h[256], in[128];
    for (int i = 0; i < 128; ++i) in[i] = 1.0f + i;
    hipMalloc(&g, 512); hipMalloc(&o, 1024);
    hipMemcpy(g, in, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, g, o, 512);
    hipMemcpy(h, o, 1024, hipMemcpyDeviceToHost);
    printf("in-range lane 0: %g %g %g %g | lane 31: %g\n", h[0], h[1], h[2], h[3], h[31 * 4]);
    printf("out-of-range lane 32: %g %g %g %g | lane 63: %g\n", h[128], h[129], h[130], h[131], h[63 * 4]);
    return 0;
}

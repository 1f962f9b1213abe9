// Microbenchmark (GPU box): throughput of v_mfma_f32_32x32x2_f32 as a function of the number of independent
// accumulator chains per wave and of waves per SIMD.   hipcc --offload-arch=gfx950 -O3 mfma_chain.hip -o mfma_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CH>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    f32x16 acc[CH];
    for (int c = 0; c < CH; ++c) for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
    float a = a0 + threadIdx.x, b = b0 + blockIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16 / CH; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CH; ++c) for (int e = 0; e < 16; ++e) s += acc[c][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int CH>
void run(float* d, int blocks_per_cu) {
    const int iters = 4000, grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<CH>, dim3(grid), dim3(256), 0, 0, d, 10, 1.f, 2.f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<CH>, dim3(grid), dim3(256), 0, 0, d, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)grid * 4 * iters * 16 * 4096.0;
    printf("chains %d waves/SIMD %d: %8.3f ms  %7.1f TFLOP/s\n", CH, blocks_per_cu, ms, flop / ms / 1e9);
}
int main() {
    float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int w = 1; w <= 4; ++w) { run<1>(d, w); run<2>(d, w); run<4>(d, w); }
    return 0;
}

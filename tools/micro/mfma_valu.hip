// Microbenchmark (GPU box): what a plain vector instruction costs a stream of dependent v_mfma_f32_32x32x2_f32 (one accumulator
// per wave, W waves per SIMD): NV `v_mov`-class instructions (or NL ds_read_b128, or NS s_nop-free scalar adds) after every group of 4
// MFMAs.   hipcc --offload-arch=gfx950 -O3 mfma_valu.hip -o mfma_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NV, int NL, int NS>
__global__ __launch_bounds__(1024) void k(float* out, int iters, float a0, float b0) {
    __shared__ float4 lds[2048];
    f32x16 acc;
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    float a = a0 + threadIdx.x, b = b0 + blockIdx.x;
    float v[8] = {a, b, a, b, a, b, a, b};
    float4 l[4]; for (int q = 0; q < 4; ++q) l[q] = make_float4(0, 0, 0, 0);
    int sacc = iters;
    lds[threadIdx.x] = make_float4(a, b, a, b); lds[threadIdx.x + 1024] = make_float4(b, a, b, a);
    __syncthreads();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < NV; ++q) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[q & 7]) : "v"(b));
#pragma unroll
            for (int q = 0; q < NL; ++q) asm volatile("ds_read_b128 %0, %1" : "=v"(l[q & 3]) : "v"((threadIdx.x * 16 + q * 1024) & 32767));
#pragma unroll
            for (int q = 0; q < NS; ++q) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc));
            __builtin_amdgcn_sched_barrier(0);
        }
        if (NL) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float s = sacc;
    for (int e = 0; e < 16; ++e) s += acc[e];
    for (int q = 0; q < 8; ++q) s += v[q];
    for (int q = 0; q < 4; ++q) s += l[q].x + l[q].y + l[q].z + l[q].w;
    out[blockIdx.x * 1024 + threadIdx.x] = s;
}
template <int NV, int NL, int NS>
void run(float* d, int waves_per_simd) {
    const int iters = 1500, grid = 256, threads = 256 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NV, NL, NS>), dim3(grid), dim3(threads), 0, 0, d, 10, 1.f, 2.f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV, NL, NS>), dim3(grid), dim3(threads), 0, 0, d, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double nm = (double)iters * 16 * waves_per_simd;            // MFMAs per SIMD
    double flop = (double)grid * 4 * nm * 4096.0;
    printf("waves/SIMD %d  per 4 MFMAs: %2d valu %d ds_read_b128 %2d salu: %8.3f ms  %6.1f TFLOP/s  %5.1f ns per MFMA and SIMD (64 cycles at 2.4 GHz = 26.7)\n",
           waves_per_simd, NV, NL, NS, ms, flop / ms / 1e9, ms * 1e6 / nm);
}
int main() {
    float* d; hipMalloc(&d, 256 * 1024 * 4);
    for (int w = 1; w <= 4; w *= 2) {
        run<0, 0, 0>(d, w); run<2, 0, 0>(d, w); run<4, 0, 0>(d, w); run<8, 0, 0>(d, w); run<16, 0, 0>(d, w); run<32, 0, 0>(d, w);
        run<0, 2, 0>(d, w); run<0, 4, 0>(d, w); run<0, 0, 8>(d, w); run<0, 0, 16>(d, w); run<8, 2, 4>(d, w);
    }
    return 0;
}

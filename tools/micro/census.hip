// Microbenchmark (GPU box): how the dispatcher places a grid of LDS-heavy 256-thread workgroups on CUs.
// Each block records (xcc, se, cu) and spins; prints the histogram of blocks per CU at several grid sizes.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <map>
#include <vector>
template <int LDSB>
__global__ __launch_bounds__(256) void k(unsigned* out, long long spin) {
    __shared__ float lds[LDSB / 4];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    unsigned hw = __builtin_amdgcn_s_getreg(63492), xcc = __builtin_amdgcn_s_getreg(63508);
    long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < spin) { __builtin_amdgcn_s_sleep(8); }
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc + (unsigned)lds[5] * 0; }
}
template <int LDSB>
void run(int grid) {
    unsigned* d; hipMalloc(&d, grid * 8);
    hipLaunchKernelGGL(k<LDSB>, dim3(grid), dim3(256), 0, 0, d, 200000LL);
    hipDeviceSynchronize();
    std::vector<unsigned> h(2 * grid); hipMemcpy(h.data(), d, grid * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, int> cnt;
    for (int b = 0; b < grid; ++b) {
        unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
        unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        cnt[(xcc << 12) | (se << 8) | (sh << 4) | cu]++;
    }
    int hist[16] = {0};
    for (auto& kv : cnt) hist[kv.second < 15 ? kv.second : 15]++;
    printf("LDS %6d B grid %5d: distinct CUs %3zu; blocks/CU histogram:", LDSB, grid, cnt.size());
    for (int i = 1; i < 10; ++i) if (hist[i]) printf("  %dx:%d", i, hist[i]);
    printf("\n");
    hipFree(d);
}
int main() {
    for (int g : {256, 384, 512, 760, 1008}) run<36880>(g);
    for (int g : {256, 384, 512, 760}) run<55296>(g);
    return 0;
}

// Batch assembly and small utility kernels, gfx950 (reference jamie.py:552-604).
#include "common.h"

thread_local char g_jamie_err[512] = {0};

extern "C" const char* jamie_last_error(void) { return g_jamie_err; }
extern "C" int jamie_version(void) { return 100; }
extern "C" int jamie_max_partials(void) { return JAMIE_MAX_PARTIALS; }

// ---- dst[b,:] = src[idx[b],:]  (dataset[i][random_batch[i]], jamie.py:583): one wave per row piece ----
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, long long n_rows, int d,
                                                          const int32_t* __restrict__ idx, int B,
                                                          float* __restrict__ dst, int vec) {
    const int b = blockIdx.y;
    long long r = idx[b];
    if (r < 0) r = 0;
    if (r >= n_rows) r = n_rows - 1;
    const float* s = src + r * d;
    float* o = dst + (long long)b * d;
    if (vec) {
        const int d4 = d >> 2;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < d4; i += gridDim.x * 256)
            reinterpret_cast<float4*>(o)[i] = reinterpret_cast<const float4*>(s)[i];
    } else {
        for (int i = blockIdx.x * 256 + threadIdx.x; i < d; i += gridDim.x * 256) o[i] = s[i];
    }
}

extern "C" int jamie_gather_rows(const float* src, long long n_rows, int d, const int32_t* idx, int B, float* dst,
                                 void* stream) {
    JAMIE_ARG(src && idx && dst && n_rows > 0 && d > 0 && B > 0, "null pointer / empty");
    const int vec = (d % 4 == 0) && ((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 16 == 0);
    int gx = (d / 4 + 255) / 256;
    if (gx < 1) gx = 1;
    if (gx > 8) gx = 8;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, src, n_rows, d, idx, B,
                       dst, vec);
    return jamie_launch_status("jamie_gather_rows");
}

// ---- device-side np.random.choice(N, B, replace): one workgroup, deterministic given (seed, step) ----
// Without replacement: rounds of "every unresolved slot draws; the lowest slot id wins a value; losers
// redraw".  The procedure is symmetric under relabelling of values, so the result is a uniform B-subset
// in uniformly random order.  Every wave leaves the loop: a round either resolves >= 1 slot or all are done.
#define SMP_HASH 4096
__global__ __launch_bounds__(1024) void sample_kernel(int32_t* idx, int B, long long N, long long offset, int replace,
                                                      const uint64_t* rng, int rng_stream) {
    __shared__ long long key[SMP_HASH];
    __shared__ int owner[SMP_HASH];
    __shared__ int pending;
    const int tid = threadIdx.x;
    for (int base = 0; base < B; base += 1024) {   // chunks of 1024 slots share the table across chunks
        if (base == 0)
            for (int i = tid; i < SMP_HASH; i += 1024) { key[i] = -1; owner[i] = 0x7fffffff; }
        __syncthreads();
        const int slot = base + tid;
        bool need = slot < B;
        long long val = 0;
        for (unsigned round = 0; round < 64; ++round) {
            if (tid == 0) pending = 0;
            __syncthreads();
            int pos = -1;
            if (need) {
                Philox4 r = jamie_rand4(rng, (uint32_t)rng_stream, ((uint64_t)slot << 8) | round);
                const uint64_t u = ((uint64_t)r.v[0] << 32) | r.v[1];
                val = (long long)(u % (uint64_t)N);
                if (replace) {
                    need = false;
                } else {
                    // open addressing keyed by value; claim by lowest slot id
                    unsigned hsh = (unsigned)((uint64_t)val * 0x9E3779B97F4A7C15ull >> 52) & (SMP_HASH - 1);
                    for (int probe = 0; probe < SMP_HASH; ++probe) {
                        const long long prev = (long long)atomicCAS((unsigned long long*)&key[hsh],
                                                                    (unsigned long long)-1LL, (unsigned long long)val);
                        if (prev == -1 || prev == val) { pos = (int)hsh; break; }
                        hsh = (hsh + 1) & (SMP_HASH - 1);
                    }
                    if (pos >= 0) atomicMin(&owner[pos], slot);
                }
            }
            __syncthreads();
            if (need && !replace) {
                if (pos >= 0 && owner[pos] == slot) need = false;
                else atomicAdd(&pending, 1);
            }
            __syncthreads();
            const int pend = pending;
            __syncthreads();
            if (pend == 0) break;
        }
        if (slot < B) idx[slot] = (int32_t)(val + offset);
        __syncthreads();
    }
}

extern "C" int jamie_sample_indices(int32_t* idx, int B, long long N, long long offset, int replace,
                                    const uint64_t* rng, int rng_stream, void* stream) {
    JAMIE_ARG(idx && rng && B > 0 && N > 0, "null pointer / empty");
    JAMIE_ARG(replace || (B <= N && B <= SMP_HASH / 2), "without replacement: B <= N and B <= 2048");
    JAMIE_ARG(N + offset <= 0x7fffffffLL, "indices must fit int32");
    hipLaunchKernelGGL(sample_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, idx, B, N, offset, replace, rng,
                       rng_stream);
    return jamie_launch_status("jamie_sample_indices");
}

// ---- corr[a,b] = (idx0[a] == idx1[b]) / max(1, #matches in row a)  (jamie.py:586-589 with P = I) ----
__global__ __launch_bounds__(256) void corr_from_idx_kernel(const int32_t* idx0, const int32_t* idx1, int B, float* corr) {
    __shared__ float red[4];
    const int a = blockIdx.x;
    const int ia = idx0[a];
    float cnt = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) cnt += (idx1[b] == ia) ? 1.f : 0.f;
    cnt = block_sum(cnt, red);
    const float inv = cnt > 0.f ? 1.f / cnt : 1.f;
    for (int b = threadIdx.x; b < B; b += 256) corr[(long long)a * B + b] = (idx1[b] == ia) ? inv : 0.f;
}

extern "C" int jamie_corr_from_indices(const int32_t* idx0, const int32_t* idx1, int B, float* corr, void* stream) {
    JAMIE_ARG(idx0 && idx1 && corr && B > 0, "null pointer / empty");
    hipLaunchKernelGGL(corr_from_idx_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, idx0, idx1, B, corr);
    return jamie_launch_status("jamie_corr_from_indices");
}

// ---- out[n] (+)= sum_m sum_slabs X[m,n]: 16 columns x 16 row phases per workgroup ----
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, int M, int N, int ld, int nslab,
                                                     long long slab_stride, float* out, int accumulate) {
    __shared__ float sh[16][17];
    const int c = threadIdx.x & 15, rp = threadIdx.x >> 4;
    const int col = blockIdx.x * 16 + c;
    float acc = 0.f;
    if (col < N)
        for (int s = 0; s < nslab; ++s)
            for (int m = rp; m < M; m += 16) acc += X[s * slab_stride + (long long)m * ld + col];
    sh[rp][c] = acc;
    __syncthreads();
    if (rp == 0 && col < N) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += sh[i][c];
        out[col] = accumulate ? out[col] + t : t;
    }
}

extern "C" int jamie_colsum(const float* X, int M, int N, int ld, int nslab, long long slab_stride, float* out,
                            int accumulate, void* stream) {
    JAMIE_ARG(X && out && M > 0 && N > 0 && ld >= N && nslab >= 1, "null pointer / empty");
    hipLaunchKernelGGL(colsum_kernel, dim3((N + 15) / 16), dim3(256), 0, (hipStream_t)stream, X, M, N, ld, nslab,
                       slab_stride, out, accumulate);
    return jamie_launch_status("jamie_colsum");
}

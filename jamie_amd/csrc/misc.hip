// Batch assembly and small utility kernels, gfx950 (reference jamie.py:552-604).
#include "common.h"
#include "sampler.h"
#include "colsum.h"

thread_local char g_jamie_err[512] = {0};

extern "C" const char* jamie_last_error(void) { return g_jamie_err; }
extern "C" int jamie_version(void) { return 100; }
extern "C" int jamie_panel_width(void) { return JAMIE_PANEL; }
extern "C" int jamie_max_partials(void) { return JAMIE_MAX_PARTIALS; }
extern "C" int jamie_max_norm_partials(void) { return JAMIE_MAX_NORM_PARTIALS; }

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

// ---- device-side np.random.choice(N, B, replace): sampler.h (one workgroup, deterministic given (seed, step)) ----
__global__ __launch_bounds__(1024) void sample_kernel(int32_t* idx, int B, long long N, long long offset, int replace,
                                                      const uint64_t* rng, int rng_stream) {
    __shared__ SampleLds lds;
    jamie_sample_block(lds, idx, B, N, offset, replace, rng, rng_stream);
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

// several independent draws in ONE launch, one workgroup each (the hybrid sampler of partial-correspondence training draws pair
// numbers and rows of both modalities: three one-workgroup launches before)
struct SampleGroup { SampleArgs a[4]; };
__global__ __launch_bounds__(1024) void sample_group_kernel(SampleGroup g, const uint64_t* rng) {
    __shared__ SampleLds lds;
    const SampleArgs& s = g.a[blockIdx.x];
    jamie_sample_block(lds, s.idx, s.B, s.N, s.offset, s.replace, rng, s.rng_stream, s.step_add);
}

extern "C" int jamie_sample_indices_group(const jamie_sample_args* list, int count, const uint64_t* rng, void* stream) {
    JAMIE_ARG(list && rng && count >= 1 && count <= 4, "1 <= count <= 4");
    SampleGroup g;
    memset(&g, 0, sizeof(g));
    for (int i = 0; i < count; ++i) {
        const jamie_sample_args& s = list[i];
        JAMIE_ARG(s.idx && s.B > 0 && s.N > 0, "null pointer / empty");
        JAMIE_ARG(s.replace || (s.B <= s.N && s.B <= SMP_HASH / 2), "without replacement: B <= N and B <= 2048");
        JAMIE_ARG(s.N + s.offset <= 0x7fffffffLL, "indices must fit int32");
        g.a[i].idx = s.idx; g.a[i].B = s.B; g.a[i].N = s.N; g.a[i].offset = s.offset; g.a[i].replace = s.replace;
        g.a[i].rng_stream = s.rng_stream; g.a[i].step_add = s.step_add;
    }
    hipLaunchKernelGGL(sample_group_kernel, dim3(count), dim3(1024), 0, (hipStream_t)stream, g, rng);
    return jamie_launch_status("jamie_sample_indices_group");
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

// ---- blk[a,b] = P[idx0[a] + row_off, idx1[b] + col_off] of a CSR matrix, optionally row-normalised (a zero row keeps
// divisor 1): the P block of jamie.py:586-589 for a sparse partial-correspondence matrix, without an N x N array.
// One workgroup per output row; every thread binary-searches its columns in the row's sorted column indices.
__global__ __launch_bounds__(256) void csr_block_kernel(const int32_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                                        const float* __restrict__ vals, const int32_t* __restrict__ idx0,
                                                        const int32_t* __restrict__ idx1, int B1, int row_off, int col_off,
                                                        int normalise, float* __restrict__ out) {
    __shared__ float red[4];
    const int a = blockIdx.x;
    const int r = idx0[a] + row_off;
    const int beg = indptr[r], end = indptr[r + 1];
    float sum = 0.f;
    for (int b = threadIdx.x; b < B1; b += 256) {
        const int c = idx1[b] + col_off;
        int lo = beg, hi = end;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (indices[mid] < c) lo = mid + 1; else hi = mid;
        }
        const float v = (lo < end && indices[lo] == c) ? vals[lo] : 0.f;
        out[(long long)a * B1 + b] = v;
        sum += v;
    }
    if (!normalise) return;
    sum = block_sum(sum, red);
    if (sum == 0.f) return;
    for (int b = threadIdx.x; b < B1; b += 256) out[(long long)a * B1 + b] /= sum;   // own elements: no barrier needed
}

extern "C" int jamie_csr_block(const int32_t* indptr, const int32_t* indices, const float* vals, const int32_t* idx0,
                               const int32_t* idx1, int B0, int B1, int row_off, int col_off, int normalise, float* out,
                               void* stream) {
    JAMIE_ARG(indptr && idx0 && idx1 && out && B0 > 0 && B1 > 0, "null pointer / empty");
    JAMIE_ARG(row_off >= 0 && col_off >= 0, "negative offset");
    hipLaunchKernelGGL(csr_block_kernel, dim3(B0), dim3(256), 0, (hipStream_t)stream, indptr, indices, vals, idx0, idx1,
                       B1, row_off, col_off, normalise, out);
    return jamie_launch_status("jamie_csr_block");
}

// ---- blk[a,b] = M[idx0[a] + row_off, idx1[b] + col_off] of a DENSE matrix (P or the correspondence F), optionally
// row-normalised (zero rows keep divisor 1; jamie.py:586-595), then out = w_blk * blk + w_add * add (the mix
// corr = PF_Ratio * P + (1 - PF_Ratio) * F of jamie.py:604).  One workgroup per output row: the B x B gather never
// materialises the [B, N] slab that `M[idx0][:, idx1]` builds in the reference.
__global__ __launch_bounds__(256) void dense_block_kernel(const float* __restrict__ M, long long ld, const int32_t* __restrict__ idx0,
                                                          const int32_t* __restrict__ idx1, int B1, long long row_off,
                                                          long long col_off, int normalise, float w_blk,
                                                          const float* __restrict__ add, float w_add, float* __restrict__ out) {
    __shared__ float red[4];
    const int a = blockIdx.x;
    const float* row = M + ((long long)idx0[a] + row_off) * ld + col_off;
    float sum = 0.f;
    if (normalise) {
        for (int b = threadIdx.x; b < B1; b += 256) sum += row[idx1[b]];
        sum = block_sum(sum, red);
    }
    const float div = (normalise && sum != 0.f) ? sum : 1.f;
    for (int b = threadIdx.x; b < B1; b += 256) {
        float v = (row[idx1[b]] / div) * w_blk;
        if (add) v += w_add * add[(long long)a * B1 + b];
        out[(long long)a * B1 + b] = v;
    }
}

// out = a * x + b * y (y may be NULL): the correspondence mix PF_Ratio * P + (1 - PF_Ratio) * F of jamie.py:604 on blocks that
// other kernels produced (jamie_csr_block, jamie_corr_from_indices)
__global__ __launch_bounds__(256) void axpby_kernel(float* __restrict__ out, float a, const float* __restrict__ x, float b,
                                                    const float* __restrict__ y, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = y ? a * x[i] + b * y[i] : a * x[i];
}

extern "C" int jamie_axpby(float* out, float a, const float* x, float b, const float* y, long long n, void* stream) {
    JAMIE_ARG(out && x && n > 0, "null pointer / empty");
    hipLaunchKernelGGL(axpby_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, a, x, b, y, n);
    return jamie_launch_status("jamie_axpby");
}

extern "C" int jamie_dense_block(const float* M, long long ld, const int32_t* idx0, const int32_t* idx1, int B0, int B1,
                                 long long row_off, long long col_off, int normalise, float w_blk, const float* add,
                                 float w_add, float* out, void* stream) {
    JAMIE_ARG(M && idx0 && idx1 && out && B0 > 0 && B1 > 0 && ld > 0, "null pointer / empty");
    JAMIE_ARG(row_off >= 0 && col_off >= 0, "negative offset");
    hipLaunchKernelGGL(dense_block_kernel, dim3(B0), dim3(256), 0, (hipStream_t)stream, M, ld, idx0, idx1, B1, row_off,
                       col_off, normalise, w_blk, add, w_add, out);
    return jamie_launch_status("jamie_dense_block");
}

// ---- out[n] (+)= sum_m sum_slabs X[m,n]  (colsum.h); up to 4 matrices per launch ----
__global__ __launch_bounds__(256) void colsum_kernel(ColsumGroup g) {
    __shared__ float4 sh[32][17];
    float* o;
    colsum_block(g, (int)blockIdx.x, sh, &o);
}

extern "C" int jamie_colsum_group(const jamie_colsum_problem* pr, int count, void* stream) {
    JAMIE_ARG(pr && count >= 1 && count <= JAMIE_MAX_GROUP, "1 <= count <= JAMIE_MAX_GROUP");
    ColsumGroup g;
    memset(&g, 0, sizeof(g));
    g.count = count;
    int blocks = 0;
    for (int i = 0; i < count; ++i) {
        const jamie_colsum_problem& s = pr[i];
        JAMIE_ARG(s.X && s.out && s.M > 0 && s.N > 0 && s.ld >= s.N && s.nslab >= 1, "null pointer / empty");
        ColsumDev& d = g.p[i];
        d.X = s.X; d.out = s.out; d.slab_stride = s.slab_stride; d.M = s.M; d.N = s.N; d.ld = s.ld; d.nslab = s.nslab;
        d.accumulate = s.accumulate; d.blk_begin = blocks;
        blocks += (s.N + 63) / 64;
    }
    hipLaunchKernelGGL(colsum_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g);
    return jamie_launch_status("jamie_colsum_group");
}

extern "C" int jamie_colsum(const float* X, int M, int N, int ld, int nslab, long long slab_stride, float* out,
                            int accumulate, void* stream) {
    jamie_colsum_problem p;
    p.X = X; p.out = out; p.M = M; p.N = N; p.ld = ld; p.nslab = nslab; p.slab_stride = slab_stride; p.accumulate = accumulate;
    return jamie_colsum_group(&p, 1, stream);
}

// ------------------------------------------------------------------------------------------------
// Device-side `preclass(axis=0)` (reference utilities.py:654-678, built at jamie.py:462-465): per-feature mean and
// population standard deviation of a [N, d] matrix, then (x - mean) / std with NaN -> 0, fp32 out.  The reference
// does this in fp64 numpy on the host; here the cells are uploaded once in their own dtype (fp32 or fp64), the
// statistics are accumulated in fp64 in numpy's two-pass order (mean, then mean of squared deviations) and the
// standardised fp32 matrix is written straight into the buffer the training loop gathers from.
// Deterministic: per-(row block, column) partial sums, added in a fixed order.
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void col_moment_kernel(const T* __restrict__ X, long long N, int d, long long ld,
                                                         const double* __restrict__ mean, double* partials) {
    __shared__ double sh[4][64];
    const int c = threadIdx.x & 63, rp = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + c;
    const long long rows_per = (N + gridDim.y - 1) / gridDim.y;
    const long long r0 = (long long)blockIdx.y * rows_per, r1 = min(N, r0 + rows_per);
    double acc = 0.0;
    if (col < d) {
        const double mu = mean ? mean[col] : 0.0;
        if (mean) {
            for (long long r = r0 + rp; r < r1; r += 4) { const double v = (double)X[r * ld + col] - mu; acc += v * v; }
        } else {
            for (long long r = r0 + rp; r < r1; r += 4) acc += (double)X[r * ld + col];
        }
    }
    sh[rp][c] = acc;
    __syncthreads();
    if (rp == 0 && col < d) partials[(long long)blockIdx.y * d + col] = (sh[0][c] + sh[1][c]) + (sh[2][c] + sh[3][c]);
}

__global__ __launch_bounds__(256) void col_moment_finish_kernel(const double* partials, int R, int d, double inv_n,
                                                                int take_sqrt, double* out) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= d) return;
    double s = 0.0;
    for (int r = 0; r < R; ++r) s += partials[(long long)r * d + col];
    s *= inv_n;
    out[col] = take_sqrt ? sqrt(s) : s;
}

template <typename T>
__global__ __launch_bounds__(256) void standardise_kernel(const T* __restrict__ X, long long N, int d, long long ld,
                                                          const double* __restrict__ mean, const double* __restrict__ sd,
                                                          float* __restrict__ out) {
    const long long total = N * d;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / d;
        const int c = (int)(i - r * d);
        const double v = ((double)X[r * ld + c] - mean[c]) / sd[c];          // utilities.py:663-668
        out[i] = (v != v) ? 0.f : (float)v;                                   // out[np.isnan(out)] = 0
    }
}

extern "C" int jamie_col_stats(const void* X, int is_f64, long long N, int d, long long ld, double* partials,
                               int n_row_blocks, double* mean, double* sd, void* stream) {
    JAMIE_ARG(X && partials && mean && sd && N > 0 && d > 0 && ld >= d, "null pointer / empty");
    JAMIE_ARG(n_row_blocks >= 1 && n_row_blocks <= 1024, "1 <= n_row_blocks <= 1024");
    const dim3 grid((d + 63) / 64, n_row_blocks), fin((d + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
    for (int pass = 0; pass < 2; ++pass) {
        const double* mu = pass ? mean : nullptr;
        if (is_f64) hipLaunchKernelGGL(col_moment_kernel<double>, grid, dim3(256), 0, st, (const double*)X, N, d, ld, mu, partials);
        else hipLaunchKernelGGL(col_moment_kernel<float>, grid, dim3(256), 0, st, (const float*)X, N, d, ld, mu, partials);
        hipLaunchKernelGGL(col_moment_finish_kernel, fin, dim3(256), 0, st, partials, n_row_blocks, d, 1.0 / (double)N, pass,
                           pass ? sd : mean);
    }
    return jamie_launch_status("jamie_col_stats");
}

extern "C" int jamie_standardise(const void* X, int is_f64, long long N, int d, long long ld, const double* mean,
                                 const double* sd, float* out, void* stream) {
    JAMIE_ARG(X && mean && sd && out && N > 0 && d > 0 && ld >= d, "null pointer / empty");
    long long b = (N * d + 255) / 256;
    const int grid = (int)(b > 8192 ? 8192 : b);
    if (is_f64) hipLaunchKernelGGL(standardise_kernel<double>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const double*)X, N, d, ld, mean, sd, out);
    else hipLaunchKernelGGL(standardise_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float*)X, N, d, ld, mean, sd, out);
    return jamie_launch_status("jamie_standardise");
}

// ------------------------------------------------------------------------------------------------
// Device counterpart of the 'hybrid' sampler of partial-correspondence training (reference jamie.py:559-573, with the
// correction described in jamie_amd/jamie.py): every batch slot is, with probability `true_ratio`, one of the known
// cell pairs (modality 0 gets the pair's row, modality 1 its column), otherwise an independent random cell of each
// modality.  The candidate draws (B distinct pair numbers, B distinct rows per modality) come from jamie_sample_indices;
// this kernel only picks per slot.  Philox keyed by (seed, step, stream): deterministic.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hybrid_assemble_kernel(const int32_t* __restrict__ pairs, const int32_t* __restrict__ pidx,
                                                              const int32_t* __restrict__ r0, const int32_t* __restrict__ r1,
                                                              int B, int num_corr, float true_ratio, const uint64_t* rng,
                                                              int rng_stream, int32_t* idx0, int32_t* idx1) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const Philox4 u = jamie_rand4(rng, (uint32_t)rng_stream, (uint64_t)b);
    const float x = (float)u.v[0] * 2.3283064365386963e-10f;
    const bool paired = num_corr > 0 && x < true_ratio;
    if (paired) {
        const int k = pidx[b] % num_corr;
        idx0[b] = pairs[2 * k];
        idx1[b] = pairs[2 * k + 1];
    } else {
        idx0[b] = r0[b];
        idx1[b] = r1[b];
    }
}

extern "C" int jamie_hybrid_assemble(const int32_t* pairs, const int32_t* pidx, const int32_t* r0, const int32_t* r1, int B,
                                     int num_corr, float true_ratio, const uint64_t* rng, int rng_stream, int32_t* idx0,
                                     int32_t* idx1, void* stream) {
    JAMIE_ARG(pidx && r0 && r1 && rng && idx0 && idx1 && B > 0 && (num_corr == 0 || pairs), "null pointer / empty");
    JAMIE_ARG(true_ratio >= 0.f && true_ratio <= 1.f && num_corr >= 0, "0 <= true_ratio <= 1, num_corr >= 0");
    hipLaunchKernelGGL(hybrid_assemble_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, pairs, pidx, r0, r1, B,
                       num_corr, true_ratio, rng, rng_stream, idx0, idx1);
    return jamie_launch_status("jamie_hybrid_assemble");
}

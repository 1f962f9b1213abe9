// BatchNorm1d (training statistics) + LeakyReLU + Dropout, forward and backward, gfx950.
//
// Replaces native_batch_norm / leaky_relu / bernoulli_+mul and their autograd backward
// (reference model.py:152-154,162-164,193-195,198-200; jamie.py:734).
//
// One workgroup owns a strip of CW = 16 feature columns for ALL batch rows, so the batch statistics
// (exact two-pass mean / biased variance, like ATen's CPU kernel) are local to the workgroup:
// 256 threads = 16 columns x 16 row phases; with B <= 512 every thread keeps its <= 32 values in
// registers and h is read from HBM/L2 exactly once.  Larger batches fall back to re-reading.
// The pre-BN activations may arrive as split-K slabs; they are summed on the fly and the sum is written
// back to slab 0 for the backward pass.  Dropout masks come from Philox (seed, step, stream, element)
// and are regenerated identically in the backward kernel; tests pass explicit masks.
#include "common.h"

#define BN_CW 16
#define BN_RP 16
#define BN_MAXR 32

struct BnFwdDev {
    float* h; const float* gamma; const float* beta; float* rmean; float* rvar;
    float* smean; float* sinvstd; float* out; const uint8_t* mask;
    long long slab_stride;
    int nslab, B, N, rng_stream, blk_begin;
};
struct BnFwdGroup { BnFwdDev p[JAMIE_MAX_GROUP]; int count; };

struct BnBwdDev {
    float* da; const float* h; const float* gamma; const float* beta; const float* smean;
    const float* sinvstd; float* dgamma; float* dbeta; float* dbias; const uint8_t* mask;
    long long slab_stride;
    int nslab, B, N, rng_stream, accumulate, blk_begin;
};
struct BnBwdGroup { BnBwdDev p[JAMIE_MAX_GROUP]; int count; };

// sum over the 16 row phases of one column; every thread of the column gets the total
__device__ __forceinline__ float col_reduce(float v, float (*sh)[BN_CW + 1], int rp, int c) {
    __syncthreads();
    sh[rp][c] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < BN_RP; ++i) t += sh[i][c];
    return t;
}

template <bool CACHED>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(BnFwdGroup g, float p_drop, float momentum, float eps,
                                                         float slope, const uint64_t* rng) {
    __shared__ float sh[BN_RP][BN_CW + 1];
    int pi = 0;
#pragma unroll
    for (int i = 1; i < JAMIE_MAX_GROUP; ++i)
        if (i < g.count && (int)blockIdx.x >= g.p[i].blk_begin) pi = i;
    const BnFwdDev& P = g.p[pi];
    const int c = threadIdx.x % BN_CW, rp = threadIdx.x / BN_CW;
    const int col = ((int)blockIdx.x - P.blk_begin) * BN_CW + c;
    const bool cok = col < P.N;
    const int B = P.B, N = P.N;

    auto load_h = [&](int row) -> float {
        float v = 0.f;
        const long long o = (long long)row * N + col;
        for (int s = 0; s < P.nslab; ++s) v += P.h[o + s * P.slab_stride];
        return v;
    };

    float v[CACHED ? BN_MAXR : 1];
    float sum = 0.f;
    if (CACHED) {
#pragma unroll
        for (int j = 0; j < BN_MAXR; ++j) {
            const int row = rp + j * BN_RP;
            v[j] = (cok && row < B) ? load_h(row) : 0.f;
            sum += v[j];
        }
    } else {
        if (cok)
            for (int row = rp; row < B; row += BN_RP) sum += load_h(row);
    }
    const float mean = col_reduce(sum, sh, rp, c) / (float)B;
    float sq = 0.f;
    if (CACHED) {
#pragma unroll
        for (int j = 0; j < BN_MAXR; ++j) {
            const int row = rp + j * BN_RP;
            const float d = v[j] - mean;
            if (row < B) sq += d * d;
        }
    } else {
        if (cok)
            for (int row = rp; row < B; row += BN_RP) {
                const float d = load_h(row) - mean;
                sq += d * d;
            }
    }
    const float var = col_reduce(sq, sh, rp, c) / (float)B;   // biased
    const float invstd = rsqrtf(var + eps);
    if (!cok) return;
    if (rp == 0) {
        P.smean[col] = mean;
        P.sinvstd[col] = invstd;
        const float unb = B > 1 ? var * ((float)B / (float)(B - 1)) : var;
        P.rmean[col] = (1.f - momentum) * P.rmean[col] + momentum * mean;
        P.rvar[col] = (1.f - momentum) * P.rvar[col] + momentum * unb;
    }
    const float ga = P.gamma[col], be = P.beta[col];
    const bool drop = p_drop > 0.f;
    const float keep_scale = drop ? 1.f / (1.f - p_drop) : 1.f;
    const uint32_t thr = jamie_drop_threshold(p_drop);
    auto finish = [&](int row, float hv) {
        const long long o = (long long)row * N + col;
        if (P.nslab > 1) P.h[o] = hv;
        float y = (hv - mean) * invstd * ga + be;
        y = y > 0.f ? y : slope * y;
        if (drop) {
            const bool keep = P.mask ? (P.mask[o] != 0) : jamie_keep(rng, (uint32_t)P.rng_stream, (uint64_t)o, thr);
            y = keep ? y * keep_scale : 0.f;
        }
        P.out[o] = y;
    };
    if (CACHED) {
#pragma unroll
        for (int j = 0; j < BN_MAXR; ++j) {
            const int row = rp + j * BN_RP;
            if (row < B) finish(row, v[j]);
        }
    } else {
        for (int row = rp; row < B; row += BN_RP) finish(row, load_h(row));
    }
}

template <bool CACHED>
__global__ __launch_bounds__(256) void bn_act_bwd_kernel(BnBwdGroup g, float p_drop, float slope,
                                                         const uint64_t* rng) {
    __shared__ float sh[BN_RP][BN_CW + 1];
    int pi = 0;
#pragma unroll
    for (int i = 1; i < JAMIE_MAX_GROUP; ++i)
        if (i < g.count && (int)blockIdx.x >= g.p[i].blk_begin) pi = i;
    const BnBwdDev& P = g.p[pi];
    const int c = threadIdx.x % BN_CW, rp = threadIdx.x / BN_CW;
    const int col = ((int)blockIdx.x - P.blk_begin) * BN_CW + c;
    const bool cok = col < P.N;
    const int B = P.B, N = P.N;
    const float mean = cok ? P.smean[col] : 0.f, invstd = cok ? P.sinvstd[col] : 0.f;
    const float ga = cok ? P.gamma[col] : 0.f, be = cok ? P.beta[col] : 0.f;
    const bool drop = p_drop > 0.f;
    const float keep_scale = drop ? 1.f / (1.f - p_drop) : 1.f;
    const uint32_t thr = jamie_drop_threshold(p_drop);

    // dy (grad wrt BN output) and xn (normalised input) of one element
    auto elem = [&](int row, float& dy, float& xn) {
        const long long o = (long long)row * N + col;
        float d = 0.f;
        for (int s = 0; s < P.nslab; ++s) d += P.da[o + s * P.slab_stride];
        xn = (P.h[o] - mean) * invstd;
        const float y = xn * ga + be;
        if (drop) {
            const bool keep = P.mask ? (P.mask[o] != 0) : jamie_keep(rng, (uint32_t)P.rng_stream, (uint64_t)o, thr);
            d = keep ? d * keep_scale : 0.f;
        }
        dy = y > 0.f ? d : d * slope;
    };

    float dyv[CACHED ? BN_MAXR : 1], xnv[CACHED ? BN_MAXR : 1];
    float s1 = 0.f, s2 = 0.f;
    if (CACHED) {
#pragma unroll
        for (int j = 0; j < BN_MAXR; ++j) {
            const int row = rp + j * BN_RP;
            dyv[j] = 0.f; xnv[j] = 0.f;
            if (cok && row < B) elem(row, dyv[j], xnv[j]);
            s1 += dyv[j];
            s2 += dyv[j] * xnv[j];
        }
    } else {
        if (cok)
            for (int row = rp; row < B; row += BN_RP) {
                float dy, xn;
                elem(row, dy, xn);
                s1 += dy;
                s2 += dy * xn;
            }
    }
    const float dbeta = col_reduce(s1, sh, rp, c);
    const float dgamma = col_reduce(s2, sh, rp, c);
    const float invB = 1.f / (float)B;
    const float k1 = dbeta * invB, k2 = dgamma * invB, gi = ga * invstd;
    float s3 = 0.f;
    if (CACHED) {
#pragma unroll
        for (int j = 0; j < BN_MAXR; ++j) {
            const int row = rp + j * BN_RP;
            if (cok && row < B) {
                const float dh = gi * (dyv[j] - k1 - xnv[j] * k2);
                P.da[(long long)row * N + col] = dh;
                s3 += dh;
            }
        }
    } else {
        if (cok)
            for (int row = rp; row < B; row += BN_RP) {
                float dy, xn;
                elem(row, dy, xn);
                const float dh = gi * (dy - k1 - xn * k2);
                P.da[(long long)row * N + col] = dh;   // slab 0 <- dh (slab 0 is only read by this thread)
                s3 += dh;
            }
    }
    const float dbias = col_reduce(s3, sh, rp, c);
    if (cok && rp == 0) {
        if (P.accumulate) {
            P.dgamma[col] += dgamma;
            P.dbeta[col] += dbeta;
            if (P.dbias) P.dbias[col] += dbias;
        } else {
            P.dgamma[col] = dgamma;
            P.dbeta[col] = dbeta;
            if (P.dbias) P.dbias[col] = dbias;
        }
    }
}

extern "C" int jamie_bn_act_fwd(const jamie_bnact_fwd_problem* pr, int count, float p_drop, float momentum,
                                float eps, float slope, const uint64_t* rng, void* stream) {
    JAMIE_ARG(pr && count >= 1 && count <= JAMIE_MAX_GROUP, "1 <= count <= JAMIE_MAX_GROUP");
    JAMIE_ARG(p_drop >= 0.f && p_drop < 1.f, "0 <= p < 1");
    BnFwdGroup g;
    memset(&g, 0, sizeof(g));
    g.count = count;
    int blocks = 0, maxB = 0;
    bool need_rng = false;
    for (int i = 0; i < count; ++i) {
        const jamie_bnact_fwd_problem& s = pr[i];
        JAMIE_ARG(s.h && s.gamma && s.beta && s.running_mean && s.running_var && s.save_mean && s.save_invstd && s.out,
                  "null pointer");
        JAMIE_ARG(s.B >= 1 && s.N >= 1 && s.nslab >= 1, "B, N, nslab >= 1");
        JAMIE_ARG(s.nslab == 1 || s.slab_stride >= (long long)s.B * s.N, "slab_stride too small");
        BnFwdDev& d = g.p[i];
        d.h = s.h; d.gamma = s.gamma; d.beta = s.beta; d.rmean = s.running_mean; d.rvar = s.running_var;
        d.smean = s.save_mean; d.sinvstd = s.save_invstd; d.out = s.out; d.mask = s.mask;
        d.slab_stride = s.slab_stride; d.nslab = s.nslab; d.B = s.B; d.N = s.N; d.rng_stream = s.rng_stream;
        d.blk_begin = blocks;
        blocks += (s.N + BN_CW - 1) / BN_CW;
        if (s.B > maxB) maxB = s.B;
        if (!s.mask && p_drop > 0.f) need_rng = true;
    }
    JAMIE_ARG(!need_rng || rng != nullptr, "rng state required when no explicit mask is given");
    hipStream_t st = (hipStream_t)stream;
    if (maxB <= BN_MAXR * BN_RP)
        hipLaunchKernelGGL(bn_act_fwd_kernel<true>, dim3(blocks), dim3(256), 0, st, g, p_drop, momentum, eps, slope, rng);
    else
        hipLaunchKernelGGL(bn_act_fwd_kernel<false>, dim3(blocks), dim3(256), 0, st, g, p_drop, momentum, eps, slope, rng);
    return jamie_launch_status("jamie_bn_act_fwd");
}

extern "C" int jamie_bn_act_bwd(const jamie_bnact_bwd_problem* pr, int count, float p_drop, float slope,
                                const uint64_t* rng, void* stream) {
    JAMIE_ARG(pr && count >= 1 && count <= JAMIE_MAX_GROUP, "1 <= count <= JAMIE_MAX_GROUP");
    JAMIE_ARG(p_drop >= 0.f && p_drop < 1.f, "0 <= p < 1");
    BnBwdGroup g;
    memset(&g, 0, sizeof(g));
    g.count = count;
    int blocks = 0, maxB = 0;
    bool need_rng = false;
    for (int i = 0; i < count; ++i) {
        const jamie_bnact_bwd_problem& s = pr[i];
        JAMIE_ARG(s.da && s.h && s.gamma && s.beta && s.save_mean && s.save_invstd && s.dgamma && s.dbeta,
                  "null pointer");
        JAMIE_ARG(s.B >= 1 && s.N >= 1 && s.nslab >= 1, "B, N, nslab >= 1");
        JAMIE_ARG(s.nslab == 1 || s.slab_stride >= (long long)s.B * s.N, "slab_stride too small");
        BnBwdDev& d = g.p[i];
        d.da = s.da; d.h = s.h; d.gamma = s.gamma; d.beta = s.beta; d.smean = s.save_mean;
        d.sinvstd = s.save_invstd; d.dgamma = s.dgamma; d.dbeta = s.dbeta; d.dbias = s.dbias_lin; d.mask = s.mask;
        d.slab_stride = s.slab_stride; d.nslab = s.nslab; d.B = s.B; d.N = s.N; d.rng_stream = s.rng_stream;
        d.accumulate = s.accumulate; d.blk_begin = blocks;
        blocks += (s.N + BN_CW - 1) / BN_CW;
        if (s.B > maxB) maxB = s.B;
        if (!s.mask && p_drop > 0.f) need_rng = true;
    }
    JAMIE_ARG(!need_rng || rng != nullptr, "rng state required when no explicit mask is given");
    hipStream_t st = (hipStream_t)stream;
    if (maxB <= BN_MAXR * BN_RP)
        hipLaunchKernelGGL(bn_act_bwd_kernel<true>, dim3(blocks), dim3(256), 0, st, g, p_drop, slope, rng);
    else
        hipLaunchKernelGGL(bn_act_bwd_kernel<false>, dim3(blocks), dim3(256), 0, st, g, p_drop, slope, rng);
    return jamie_launch_status("jamie_bn_act_bwd");
}

// Finalisation of the fused latent block (jamie_latent_m_*): the per-workgroup partial sums of the forward / backward
// launches -> the four losses, d sigma and the head-bias gradients.  Shared by latent.hip (the backward launch's last
// workgroup) and optim.hip (the range-norm launch's extra workgroup, when the step defers it there).
#pragma once
#include "common.h"

#define LM 4
#define LF_ROWS 32          // cells per workgroup of the fused kernels
enum { SM_MU2 = 0, SM_TROW = 4, SM_AL = 8, SM_F = 12, SM_DSIG = 13, SM_SLOTS = 17 };

struct LatFinal {
    int B, L, M, lmax, accumulate, n_rec_partials;
    const float* partials; const float* rec_partials; const float* hyper; const float* colpart;
    float* losses; float* dsigma; float* dbias_head[LM];
};

// N block-wide sums at once: wave sums on DPP (common.h), ONE pair of barriers for all N, and the cross-wave sums by N
// threads (thread k adds slot k over the waves, in wave order); the totals are left in red[0 .. N-1] for whoever needs them
// after the call's final barrier.  `red` holds (blockDim.x / 64) * N + N floats.
template <int N>
__device__ __forceinline__ void block_sum_n(float (&v)[N], float* red) {
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = wave_sum_dpp(v[k]);
    lds_barrier();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < N; ++k) red[N + w * N + k] = v[k];
    }
    lds_barrier();
    if (threadIdx.x < N) {
        float t = 0.f;
        for (int i = 0; i < nw; ++i) t += red[N + i * N + threadIdx.x];
        red[threadIdx.x] = t;
    }
    lds_barrier();
}
// `sq` (optional): receives the sum of squares of the gradients this call finalises (d sigma, head biases) -- the caller is
// then the range-norm launch, which adds it to the clip norm's partial sums; `g` / `g16`: the flat fp32 gradient and its
// optional bf16 copy (same offsets).
__device__ __forceinline__ void latent_m_finalise(const LatFinal& a, float* red, float* sq = nullptr, const float* g = nullptr,
                                                  unsigned short* g16 = nullptr) {
    const int LMAX = a.lmax;
    const int B = a.B, L = a.L, M = a.M, n = B * L, NT = blockDim.x, tid = threadIdx.x;
    const int nblk = (B + LF_ROWS - 1) / LF_ROWS;
    const float invBL = 1.f / (float)n;
    // inputs written by other CUs (cold in this CU's caches): every load is issued before the first use, so this costs
    // about one memory round trip instead of one per dependent step
    const float kl_scale = a.hyper[0], w_rec = a.hyper[1], w_al = a.hyper[2], w_f = a.hyper[3];
    const float best = a.losses[5];
    float v[SM_SLOTS + 2];
    float my_sq = 0.f;
#pragma unroll
    for (int sl = 0; sl < SM_SLOTS; ++sl)           // 17 independent loads
        v[sl] = tid < nblk ? a.partials[sl * JAMIE_MAX_PARTIALS + tid] : 0.f;
    v[SM_SLOTS] = tid < a.n_rec_partials ? a.rec_partials[tid] : 0.f;
    // head-bias gradients = column sums of d(mu | logvar): the workgroups' partial sums, added in workgroup order; the
    // first 16 partial sums of this thread's column are loaded together
    const bool col_ok = a.colpart && tid < M * 2 * L && a.dbias_head[(tid / (2 * L)) & 3] != nullptr;
    const int ci = col_ok ? tid / (2 * L) : 0, cc = col_ok ? tid % (2 * L) : 0;
    float t[16], prev = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) t[u] = (col_ok && u < nblk) ? a.colpart[((long long)u * LM + ci) * 2 * LMAX + cc] : 0.f;
    if (col_ok && a.accumulate) prev = a.dbias_head[ci][cc];
    // (rare shapes: more partial sums than the first batch)
    for (int i = tid + NT; i < nblk; i += NT)
#pragma unroll
        for (int sl = 0; sl < SM_SLOTS; ++sl) v[sl] += a.partials[sl * JAMIE_MAX_PARTIALS + i];
    for (int i = tid + NT; i < a.n_rec_partials; i += NT) v[SM_SLOTS] += a.rec_partials[i];
        if (col_ok) {
        float acc = prev;
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += t[u];
        for (int rb = 16; rb < nblk; ++rb) acc += a.colpart[((long long)rb * LM + ci) * 2 * LMAX + cc];
        a.dbias_head[ci][cc] = acc;
        my_sq += acc * acc;
        if (g16) g16[&a.dbias_head[ci][cc] - g] = __builtin_bit_cast(unsigned short, (__bf16)acc);
    }
    if (a.colpart) {            // more columns than threads: the remaining ones, the plain way
        for (int idx = tid + NT; idx < M * 2 * L; idx += NT) {
            const int i = idx / (2 * L), cidx = idx % (2 * L);
            if (!a.dbias_head[i]) continue;
            float acc = a.accumulate ? a.dbias_head[i][cidx] : 0.f;
            for (int rb = 0; rb < nblk; ++rb) acc += a.colpart[((long long)rb * LM + i) * 2 * LMAX + cidx];
            a.dbias_head[i][cidx] = acc;
            my_sq += acc * acc;
            if (g16) g16[&a.dbias_head[i][cidx] - g] = __builtin_bit_cast(unsigned short, (__bf16)acc);
        }
    }
    v[SM_SLOTS + 1] = my_sq;
    block_sum_n<SM_SLOTS + 2>(v, red);
    if (tid == 0) {
        const float rec = red[SM_SLOTS];
        float kl = 0.f, al = 0.f;
        float sq_sig = 0.f;
        for (int i = 0; i < M; ++i) {
            kl += -0.5f * (red[SM_TROW + i] / (float)L - red[SM_MU2 + i] * invBL);
            al += red[SM_AL + i];
            a.dsigma[i] = red[SM_DSIG + i];
            sq_sig += red[SM_DSIG + i] * red[SM_DSIG + i];
            if (g16) g16[&a.dsigma[i] - g] = __builtin_bit_cast(unsigned short, (__bf16)red[SM_DSIG + i]);
        }
        if (sq) *sq = red[SM_SLOTS + 1] + sq_sig;
        const float l_kl = kl_scale * kl, l_rec = w_rec * rec, l_al = w_al * al * invBL, l_f = w_f * red[SM_F] * invBL;
        const float total = l_kl + l_rec + l_al + l_f;
        a.losses[0] = l_kl; a.losses[1] = l_rec; a.losses[2] = l_al; a.losses[3] = l_f;
        a.losses[4] = total;
        a.losses[5] = fminf(best, total);
    }
}


void jamie_latent_m_fill_final(const jamie_latent_m* a, LatFinal* f);      // latent.hip

// Correspondence stage of JAMIE: the element-wise half of one Prime_Dual iteration, gfx950.
//
// Reference: JAMIE.Prime_Dual, jamie/jamie.py:314-414 (MultiOmics branch).  One iteration there is seven dense
// products plus ~25 element-wise ATen dispatches over [m, n] matrices.  Here the products are four launches of the
// fp32 MFMA GEMM (gemm_f32.hip; the scaling factor a = tr(Kx (F Ky) F^T) / tr(Kx Kx) is sum((Kx F Ky) o F) / tr(Kx Kx),
// and Kx F Ky is the product the NEXT iteration's gradient needs anyway) and everything else is three kernels:
//
//   pd_update_kernel : gradient assembly (the rank-one terms are never materialised), the Adam-style moments, the
//                      projected step and the epsilon-relaxation of F in ONE pass over F, m1, m2, G1, G2
//                      (reads 20 B, writes 12 B per element), plus per-tile partial row / column sums of the NEW F
//   pd_finish_kernel : row / column sums from the partials (fixed order: deterministic), then the slack and dual
//                      updates S, Mu, Lambda (jamie.py:386-394)
//   pd_dot_kernel / pd_alpha_kernel : a <- sum(G2 o F) / tr(Kx Kx)  (jamie.py:397-402)
//
// A wave reads 1 KiB of a row per instruction (16 B per lane); nothing is atomically accumulated.
#include "common.h"

#define PD_TR 32      // rows per tile
#define PD_TC 256     // columns per tile (64 lanes x 4)

struct PdUpdate {
    float* F; const float* G1; const float* G2; float* m1; float* m2;
    const float* Mu; const float* Lambda; const float* S; const float* rowsum; const float* colsum;
    const float* alpha;
    float* rowpart; float* colpart;
    int m, n, ncb;
    float rho, epsilon, d1, d2;
};

__global__ __launch_bounds__(256) void pd_update_kernel(PdUpdate p) {
    __shared__ float cs[4][PD_TC];
    const int lane = threadIdx.x & 63, rp = threadIdx.x >> 6;
    const int cb = blockIdx.x, rb = blockIdx.y;
    const int c0 = cb * PD_TC + lane * 4;
    const float a4 = 4.f * p.alpha[0];
    const float pho1 = 0.9f, pho2 = 0.999f, delta = 10e-8f;                    // jamie.py:347-349
    const bool vec = (p.n & 3) == 0;
    float lam[4], cterm[4], csum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = c0 + e;
        lam[e] = c < p.n ? p.Lambda[c] : 0.f;
        cterm[e] = c < p.n ? p.colsum[c] + p.S[c] - 2.f : 0.f;                 // (1^T F + (S - 2 1)^T)_j, jamie.py:366-370
    }
#pragma unroll 2
    for (int k = 0; k < PD_TR / 4; ++k) {
        const int r = rb * PD_TR + rp + 4 * k;
        float rsum = 0.f;
        if (r < p.m && c0 < p.n) {
            const long long o = (long long)r * p.n + c0;
            float f[4], g1[4], g2[4], a[4], b[4];
            if (vec) {
                const float4 vf = *reinterpret_cast<const float4*>(p.F + o), v1 = *reinterpret_cast<const float4*>(p.G1 + o);
                const float4 v2 = *reinterpret_cast<const float4*>(p.G2 + o), va = *reinterpret_cast<const float4*>(p.m1 + o);
                const float4 vb = *reinterpret_cast<const float4*>(p.m2 + o);
                f[0] = vf.x; f[1] = vf.y; f[2] = vf.z; f[3] = vf.w; g1[0] = v1.x; g1[1] = v1.y; g1[2] = v1.z; g1[3] = v1.w;
                g2[0] = v2.x; g2[1] = v2.y; g2[2] = v2.z; g2[3] = v2.w; a[0] = va.x; a[1] = va.y; a[2] = va.z; a[3] = va.w;
                b[0] = vb.x; b[1] = vb.y; b[2] = vb.z; b[3] = vb.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool ok = c0 + e < p.n;
                    f[e] = ok ? p.F[o + e] : 0.f; g1[e] = ok ? p.G1[o + e] : 0.f; g2[e] = ok ? p.G2[o + e] : 0.f;
                    a[e] = ok ? p.m1[o + e] : 0.f; b[e] = ok ? p.m2[o + e] : 0.f;
                }
            }
            const float rterm = p.Mu[r] + p.rho * p.rowsum[r];                // Mu 1^T + rho F 1 1^T, jamie.py:361,364
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float grad = 4.f * g1[e] - a4 * g2[e] + rterm + lam[e] + p.rho * cterm[e];   // jamie.py:358-373
                a[e] = pho1 * a[e] + (1.f - pho1) * grad;                                           // jamie.py:376-377
                b[e] = pho2 * b[e] + (1.f - pho2) * grad * grad;
                const float step = (a[e] / p.d1) / (sqrtf(b[e] / p.d2) + delta);                    // jamie.py:378-380
                const float ft = fmaxf(f[e] - step, 0.f);                                           // jamie.py:381-382
                f[e] = (1.f - p.epsilon) * f[e] + p.epsilon * ft;                                   // jamie.py:384
            }
            if (vec) {
                *reinterpret_cast<float4*>(p.F + o) = make_float4(f[0], f[1], f[2], f[3]);
                *reinterpret_cast<float4*>(p.m1 + o) = make_float4(a[0], a[1], a[2], a[3]);
                *reinterpret_cast<float4*>(p.m2 + o) = make_float4(b[0], b[1], b[2], b[3]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (c0 + e < p.n) { p.F[o + e] = f[e]; p.m1[o + e] = a[e]; p.m2[o + e] = b[e]; }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (c0 + e < p.n) { csum[e] += f[e]; rsum += f[e]; }
        }
        rsum = wave_sum(rsum);                                                 // this wave owns the row's 256 columns
        if (lane == 0 && r < p.m) p.rowpart[(long long)r * p.ncb + cb] = rsum;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) cs[rp][lane * 4 + e] = csum[e];
    __syncthreads();
    const int c = cb * PD_TC + threadIdx.x;
    if (c < p.n)
        p.colpart[(long long)rb * p.n + c] = (cs[0][threadIdx.x] + cs[1][threadIdx.x]) + (cs[2][threadIdx.x] + cs[3][threadIdx.x]);
}

struct PdFinish {
    const float* rowpart; const float* colpart; float* rowsum; float* colsum;
    float* S; float* Mu; float* Lambda;
    int m, n, ncb, nrb;
    float rho, epsilon;
};

// threads [0, n): columns (S, Lambda); threads [n, n + m): rows (Mu)
__global__ __launch_bounds__(256) void pd_finish_kernel(PdFinish p) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t < p.n) {
        float cs = 0.f;
        for (int rb = 0; rb < p.nrb; ++rb) cs += p.colpart[(long long)rb * p.n + t];
        p.colsum[t] = cs;
        const float lam = p.Lambda[t];
        float s = p.S[t];
        const float grad_s = lam + p.rho * (cs - 1.f + s);                     // jamie.py:387
        const float st = fmaxf(s - grad_s, 0.f);                               // jamie.py:388-389
        s = (1.f - p.epsilon) * s + p.epsilon * st;                            // jamie.py:390
        p.S[t] = s;
        p.Lambda[t] = lam + p.epsilon * (cs - 1.f + s);                        // jamie.py:394
    } else if (t < (long long)p.n + p.m) {
        const long long r = t - p.n;
        float rs = 0.f;
        for (int cb = 0; cb < p.ncb; ++cb) rs += p.rowpart[r * p.ncb + cb];
        p.rowsum[r] = rs;
        p.Mu[r] = p.Mu[r] + p.epsilon * (rs - 1.f);                           // jamie.py:393
    }
}

__global__ __launch_bounds__(256) void pd_dot_kernel(const float* __restrict__ x, const float* __restrict__ y, long long n,
                                                     float* partials) {
    __shared__ float red[4];
    const long long n4 = n >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    const float4* y4 = reinterpret_cast<const float4*>(y);
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 a = x4[i], b = y4[i];
        acc += (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
    }
    if (blockIdx.x == 0) {
        const long long i = (n4 << 2) + threadIdx.x;
        if (i < n) acc += x[i] * y[i];
    }
    const float t = block_sum(acc, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void pd_alpha_kernel(const float* partials, int n_partials, float inv_trkk, float* alpha) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n_partials; i += 256) s += partials[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) alpha[0] = s * inv_trkk;
}

extern "C" int jamie_pd_workspace(int m, int n, long long* rowpart_elems, long long* colpart_elems) {
    if (m <= 0 || n <= 0 || !rowpart_elems || !colpart_elems) return jamie_fail(-1, "%s: bad arguments [%lld %lld]", "jamie_pd_workspace", m, n);
    *rowpart_elems = (long long)m * ((n + PD_TC - 1) / PD_TC);
    *colpart_elems = (long long)((m + PD_TR - 1) / PD_TR) * n;
    return 0;
}

extern "C" int jamie_pd_step(const jamie_pd_state* s, int iteration, void* stream) {
    JAMIE_ARG(s != nullptr, "null state");
    JAMIE_ARG(s->F && s->G1 && s->G2 && s->m1 && s->m2 && s->Mu && s->Lambda && s->S && s->rowsum && s->colsum &&
                  s->alpha && s->rowpart && s->colpart, "null pointer");
    JAMIE_ARG(s->m > 0 && s->n > 0 && iteration >= 1, "empty problem / iteration < 1");
    JAMIE_ARG(((uintptr_t)s->F % 16) == 0 && ((uintptr_t)s->G1 % 16) == 0 && ((uintptr_t)s->G2 % 16) == 0 &&
                  ((uintptr_t)s->m1 % 16) == 0 && ((uintptr_t)s->m2 % 16) == 0, "matrices must be 16-byte aligned");
    const int ncb = (s->n + PD_TC - 1) / PD_TC, nrb = (s->m + PD_TR - 1) / PD_TR;
    JAMIE_ARG(nrb <= 65535, "m too large for one launch");
    PdUpdate u;
    u.F = s->F; u.G1 = s->G1; u.G2 = s->G2; u.m1 = s->m1; u.m2 = s->m2; u.Mu = s->Mu; u.Lambda = s->Lambda; u.S = s->S;
    u.rowsum = s->rowsum; u.colsum = s->colsum; u.alpha = s->alpha; u.rowpart = s->rowpart; u.colpart = s->colpart;
    u.m = s->m; u.n = s->n; u.ncb = ncb; u.rho = s->rho; u.epsilon = s->epsilon;
    u.d1 = (float)(1.0 - pow(0.9, (double)iteration));                          // 1 - pho1^i, jamie.py:378 (np.power, double)
    u.d2 = (float)(1.0 - pow(0.999, (double)iteration));
    hipLaunchKernelGGL(pd_update_kernel, dim3(ncb, nrb), dim3(256), 0, (hipStream_t)stream, u);
    PdFinish f;
    f.rowpart = s->rowpart; f.colpart = s->colpart; f.rowsum = s->rowsum; f.colsum = s->colsum;
    f.S = s->S; f.Mu = s->Mu; f.Lambda = s->Lambda; f.m = s->m; f.n = s->n; f.ncb = ncb; f.nrb = nrb;
    f.rho = s->rho; f.epsilon = s->epsilon;
    const long long tt = (long long)s->m + s->n;
    hipLaunchKernelGGL(pd_finish_kernel, dim3((unsigned)((tt + 255) / 256)), dim3(256), 0, (hipStream_t)stream, f);
    return jamie_launch_status("jamie_pd_step");
}

extern "C" int jamie_pd_alpha(const float* G2, const float* F, long long count, float* partials, int n_partials,
                              float inv_trkk, float* alpha, void* stream) {
    JAMIE_ARG(G2 && F && partials && alpha && count > 0, "null pointer / empty");
    JAMIE_ARG(((uintptr_t)G2 % 16) == 0 && ((uintptr_t)F % 16) == 0, "matrices must be 16-byte aligned");
    JAMIE_ARG(n_partials >= 1 && n_partials <= JAMIE_MAX_PARTIALS, "n_partials");
    long long b = (count / 4 + 255) / 256;
    const int grid = (int)(b < 1 ? 1 : (b > n_partials ? n_partials : b));
    hipLaunchKernelGGL(pd_dot_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, G2, F, count, partials);
    hipLaunchKernelGGL(pd_alpha_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, grid, inv_trkk, alpha);
    return jamie_launch_status("jamie_pd_alpha");
}

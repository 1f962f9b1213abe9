// Latent block of JAMIE's coupled VAE for two modalities, gfx950:
//   reparameterisation  z = mu + eps * (exp(logvar/2) + 1e-7)                 (reference model.py:225-243)
//   sigma-weighted combine with the B x B correspondence block `corr`          (model.py:245-259)
//   KL (with the reference's logvars[i] = ROW i quirk), alignment ("CosSim") and F losses and their
//   gradients w.r.t. mu, logvar, sigma                                         (jamie.py:618-668, 723-734)
//
// Everything here is [B, L] sized (B = batch, L = latent): HBM-resident but tiny, so the kernels are
// thread-per-element with wavefront-shuffle + LDS block reductions; loss terms are written as per-block
// partial sums and added up in a fixed order by block 0 of the last kernel (deterministic, no atomics).
// The B x B products (C z, C^T z, F comb, ...) are ~8 MFLOP each: plain VALU loops, not MFMA.
#include "common.h"
#include "latent_final.h"
#include "sampler.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define LAT_SLOTS 16
enum { S_MU2_0 = 0, S_MU2_1, S_TROW0, S_TROW1, S_AL0, S_AL1, S_F, S_DSIG0, S_DSIG1 };

struct LatDev {
    int B, L;
    const float* ml[2]; int ml_nslab; long long ml_slab_stride;
    const float* head_bias[2]; const float* eps_in[2];
    const float* sigma; const float* corr; const float* Fblk; const float* hyper;
    float* mu[2]; float* lv[2]; float* z[2]; float* eps[2]; float* comb[2]; float* cz[2];
    float* rsum; float* qsum; float* fc1; float* partials;
    const float* dcomb[2]; int dcomb_nslab; long long dcomb_slab_stride;
    float* H[2]; float* ch[2]; float* fte; float* dml[2]; float* dsigma;
    const float* rec_partials; int n_rec_partials; float* losses;
    int cosine; int rng_stream;
    const float* dz_ext[2]; const float* dmu_ext[2]; const float* dlv_ext;
    unsigned short* comb_bf[2]; unsigned short* combT_bf[2]; unsigned short* dml_bf[2]; unsigned short* dmlT_bf[2];
    int ident;      // corr == identity (NULL): corr z_j = z_j and the row / column sums are 1 -- no [B,B] x [B,L] launches
};

__device__ __forceinline__ unsigned short to_bf16(float v) { return __builtin_bit_cast(unsigned short, (__bf16)v); }

static LatDev to_dev(const jamie_latent* a) {
    LatDev d;
    memset(&d, 0, sizeof(d));
    d.B = a->B; d.L = a->L;
    for (int i = 0; i < 2; ++i) {
        d.ml[i] = a->ml[i]; d.head_bias[i] = a->head_bias[i]; d.eps_in[i] = a->eps_in[i];
        d.mu[i] = a->mu[i]; d.lv[i] = a->lv[i]; d.z[i] = a->z[i]; d.eps[i] = a->eps[i];
        d.comb[i] = a->comb[i]; d.cz[i] = a->cz[i]; d.dcomb[i] = a->dcomb[i]; d.H[i] = a->H[i];
        d.ch[i] = a->ch[i]; d.dml[i] = a->dml[i];
    }
    d.ml_nslab = a->ml_nslab; d.ml_slab_stride = a->ml_slab_stride;
    d.sigma = a->sigma; d.corr = a->corr; d.Fblk = a->Fblk; d.hyper = a->hyper;
    d.rsum = a->rsum; d.qsum = a->qsum; d.fc1 = a->fc1; d.partials = a->partials;
    d.dcomb_nslab = a->dcomb_nslab; d.dcomb_slab_stride = a->dcomb_slab_stride;
    d.fte = a->fte; d.dsigma = a->dsigma;
    d.rec_partials = a->rec_partials; d.n_rec_partials = a->n_rec_partials; d.losses = a->losses;
    d.cosine = a->cosine; d.rng_stream = a->rng_stream;
    d.ident = a->corr == nullptr;
    if (d.ident) { d.cz[0] = a->z[1]; d.cz[1] = a->z[0]; }      // corr z_1 = z_1, corr^T z_0 = z_0: read z itself
    for (int i = 0; i < 2; ++i) { d.dz_ext[i] = a->dz_ext[i]; d.dmu_ext[i] = a->dmu_ext[i]; }
    d.dlv_ext = a->dlv_ext;
    for (int i = 0; i < 2; ++i) {
        d.comb_bf[i] = (unsigned short*)a->comb_bf16[i]; d.combT_bf[i] = (unsigned short*)a->combT_bf16[i];
        d.dml_bf[i] = (unsigned short*)a->dml_bf16[i]; d.dmlT_bf[i] = (unsigned short*)a->dmlT_bf16[i];
    }
    return d;
}

__device__ __forceinline__ void put_partial(const LatDev& a, int slot, float v, float* red) {
    const float t = block_sum(v, red);
    if (threadIdx.x == 0) a.partials[slot * JAMIE_MAX_PARTIALS + blockIdx.x] = t;
}
// N per-workgroup partial sums behind ONE pair of barriers (block_sum_n: the same adds as N block_sum calls, bit for bit)
template <int N>
__device__ __forceinline__ void put_partials(const LatDev& a, const int (&slots)[N], float (&v)[N]) {
    __shared__ float redn[(256 / 64 + 1) * N];
    block_sum_n<N>(v, redn);
    if ((int)threadIdx.x < N) a.partials[slots[threadIdx.x] * JAMIE_MAX_PARTIALS + blockIdx.x] = redn[threadIdx.x];
}

// ---- K1: mu/logvar from the heads GEMM slabs, reparameterise, KL partial sums ----
__global__ __launch_bounds__(256) void latent_reparam_kernel(LatDev a, const uint64_t* rng) {
    __shared__ float red[4];
    const int B = a.B, L = a.L, n = B * L;
    const int e = blockIdx.x * 256 + threadIdx.x;
    const bool ok = e < n;
    const int b = ok ? e / L : 0, l = ok ? e % L : 0;
    float mu2[2] = {0.f, 0.f}, trow[2] = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (!ok) continue;
        float mu = a.head_bias[i][l], lv = a.head_bias[i][L + l];
        for (int s0 = 0; s0 < a.ml_nslab; s0 += 8) {        // eight slabs' loads in flight per round trip, added in slab order
            float tm[8], tl[8];                              // (one slab per iteration was a memory round trip each: 16 per launch)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float* p = a.ml[i] + (s0 + u) * a.ml_slab_stride + (long long)b * 2 * L;
                tm[u] = s0 + u < a.ml_nslab ? p[l] : 0.f;
                tl[u] = s0 + u < a.ml_nslab ? p[L + l] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { mu += tm[u]; lv += tl[u]; }
        }
        float ep;
        if (a.eps_in[i]) {
            ep = a.eps_in[i][e];
        } else {
            Philox4 r = jamie_rand4(rng, (uint32_t)(a.rng_stream + i), (uint64_t)e);
            float n1;
            jamie_box_muller(r.v[0], r.v[1], ep, n1);
        }
        const float sd = expf(0.5f * lv) + 1e-7f;
        a.mu[i][e] = mu; a.lv[i][e] = lv; a.eps[i][e] = ep;
        a.z[i][e] = mu + ep * sd;
        mu2[i] = mu * mu;
        // KL quirk: logvars[j] is ROW j of the LAST modality's logvar (jamie.py:619-628, model.py:243)
        if (i == 1 && b < 2) trow[b] = 1.f + lv - expf(lv);
    }
    const int slots[4] = {S_MU2_0, S_MU2_1, S_TROW0, S_TROW1};
    float pv[4] = {mu2[0], mu2[1], trow[0], trow[1]};
    put_partials<4>(a, slots, pv);
}

// ---- small [B,B] x [B,L] products.  job 0: out = Mtx Z (+ row sums); job 1: out = Mtx^T Z (+ column sums).
// Mtx == nullptr means identity (out = Z, sums = 1). ----
struct MmJob { const float* Mtx; const float* Z; float* out; float* sums; int transpose; int active; };
struct MmJobs { MmJob j[2]; int B, L, l0; };       // l0: first latent column of this launch (L > 128: several launches)

// Exact-fp32 MFMA (v_mfma_f32_32x32x2_f32): workgroup = 32 output rows x all L columns, 16 waves; wave w multiplies the k-slice
// [w KS, (w + 1) KS) of the B columns of the matrix (operands straight from global memory in the MFMA lane layout: 4 bytes per
// lane and k-step, 8-16 k-steps' loads in flight together), the 16 partial tiles are added in wave order through LDS (wave w
// adds accumulator register w).  The row
// (job 1: column) sums of the matrix ride along.  History: one thread per output walking a matrix row from global memory (1024
// dependent round trips, 150-290 us per launch at B = 512-1024); LDS-staged 64-column chunks on the vector ALU (16-44 us: a
// latency chain of chunk loads and LDS reads on four waves); this one 6-9 us.  The general correspondence / F blocks of
// partial-correspondence training pay four to six such products per step, config 1 (batch with duplicates) two.
template <int NCT>
__global__ __launch_bounds__(1024) void small_mm_kernel(MmJobs js) {
    __shared__ float red[16][16][64];                 // one 32 x 32 tile per wave
    __shared__ float smred[16][64];
    const MmJob& J = js.j[blockIdx.y];
    if (!J.active) return;
    const int B = js.B, L = js.L, l0 = js.l0, tid = threadIdx.x;
    const int row0 = blockIdx.x * 32;
    if (J.Mtx == nullptr) {                           // identity: out = Z, sums = 1
        if (l0 == 0) {
            for (int i = tid; i < 32 * L; i += 1024) {
                const int row = row0 + i / L, l = i % L;
                if (row < B) { J.out[(long long)row * L + l] = J.Z[(long long)row * L + l]; if (J.sums && l == 0) J.sums[row] = 1.f; }
            }
        }
        return;
    }
    const int w = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int KS = ((B + 15) / 16 + 1) / 2 * 2;       // k-slice per wave (even)
    const int kbeg = w * KS, kend = min(B, kbeg + KS);
    const int row = row0 + r;
    f32x16 acc[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[ct][e] = 0.f;
    float sm = 0.f;
    constexpr int NB = NCT == 1 ? 16 : 8;             // k-steps (of two columns each) whose loads are in flight together
    for (int k0 = kbeg; k0 < kend; k0 += 2 * NB) {
        float av[NB], bv[NB][NCT];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int kk = k0 + 2 * u + h;
            const bool ok = kk < kend;
            av[u] = (ok && row < B) ? (J.transpose ? J.Mtx[(long long)kk * B + row] : J.Mtx[(long long)row * B + kk]) : 0.f;
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const int n = l0 + ct * 32 + r;
                bv[u][ct] = (ok && n < L) ? J.Z[(long long)kk * L + n] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            sm += av[u];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u][ct], acc[ct], 0, 0, 0);
        }
    }
    smred[w][lane] = sm;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        if (ct) lds_barrier();                        // (the previous tile's sums have been read)
#pragma unroll
        for (int e = 0; e < 16; ++e) red[w][e][lane] = acc[ct][e];
        lds_barrier();
        {                                             // wave w adds accumulator register w of all 16 waves (in wave order)
            const int n = l0 + ct * 32 + r, e = w;
            float t = 0.f;
#pragma unroll
            for (int ww = 0; ww < 16; ++ww) t += red[ww][e][lane];
            const int m = row0 + (e & 3) + 8 * (e >> 2) + 4 * h;
            if (m < B && n < L) J.out[(long long)m * L + n] = t;
        }
    }
    if (J.sums && l0 == 0 && w == 1 && lane < 32 && row < B) {   // (smred was published by the barrier above)
        float t = 0.f;
#pragma unroll
        for (int ww = 0; ww < 16; ++ww) t += smred[ww][lane] + smred[ww][lane + 32];
        J.sums[row] = t;
    }
}

static void launch_mm(hipStream_t st, int B, int L, const MmJob& j0, const MmJob& j1) {
    MmJobs js;
    js.j[0] = j0; js.j[1] = j1; js.B = B; js.L = L;
    const dim3 grid((B + 31) / 32, 2);
    for (int l0 = 0; l0 < L; l0 += 128) {
        js.l0 = l0;
        const int nct = (min(L - l0, 128) + 31) / 32;
        if (nct <= 1) hipLaunchKernelGGL(small_mm_kernel<1>, grid, dim3(1024), 0, st, js);
        else if (nct == 2) hipLaunchKernelGGL(small_mm_kernel<2>, grid, dim3(1024), 0, st, js);
        else if (nct == 3) hipLaunchKernelGGL(small_mm_kernel<3>, grid, dim3(1024), 0, st, js);
        else hipLaunchKernelGGL(small_mm_kernel<4>, grid, dim3(1024), 0, st, js);
    }
}

// per-row cosine pieces: s = a.c / (|a||c|)
__device__ __forceinline__ void cos_row(const float* a, const float* c, int L, float& dot, float& na2, float& nc2) {
    dot = 0.f; na2 = 0.f; nc2 = 0.f;
    for (int l = 0; l < L; ++l) {
        dot = fmaf(a[l], c[l], dot);
        na2 = fmaf(a[l], a[l], na2);
        nc2 = fmaf(c[l], c[l], nc2);
    }
}

// ---- K3: combine + alignment partial sums ----
__global__ __launch_bounds__(256) void latent_combine_kernel(LatDev a) {
    __shared__ float red[4];
    const int B = a.B, L = a.L, n = B * L;
    const int e = blockIdx.x * 256 + threadIdx.x;
    const bool ok = e < n;
    float al0 = 0.f, al1 = 0.f;
    if (ok) {
        const int b = e / L;
        const float s0 = a.sigma[0], s1 = a.sigma[1];
        const float z0 = a.z[0][e], z1 = a.z[1][e];
        const float c0 = (s0 * z0 + s1 * a.cz[0][e]) / (s0 + s1 * (a.ident ? 1.f : a.rsum[b]));
        const float c1 = (s1 * z1 + s0 * a.cz[1][e]) / (s1 + s0 * (a.ident ? 1.f : a.qsum[b]));
        a.comb[0][e] = c0;
        a.comb[1][e] = c1;
        // bf16 compute mode: the decoder's first GEMM and its dW read bf16 [B,L] / [L,B] copies (tiny: B*L elements)
        const int l = e % L;
        if (a.comb_bf[0]) { a.comb_bf[0][e] = to_bf16(c0); a.comb_bf[1][e] = to_bf16(c1); }
        if (a.combT_bf[0]) { a.combT_bf[0][(long long)l * B + b] = to_bf16(c0); a.combT_bf[1][(long long)l * B + b] = to_bf16(c1); }
        if (!a.cosine) {
            al0 = (z0 - c0) * (z0 - c0);
            al1 = (z1 - c1) * (z1 - c1);
        }
    }
    if (!a.cosine) {
        put_partial(a, S_AL0, al0, red);
        put_partial(a, S_AL1, al1, red);
    }
}

// cosine alignment: per-row (1 - cos)^2, one thread per row (L is small)
__global__ __launch_bounds__(256) void latent_cosine_loss_kernel(LatDev a) {
    __shared__ float red[4];
    const int b = blockIdx.x * 256 + threadIdx.x;
    float v[2] = {0.f, 0.f};
    if (b < a.B) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float dot, na2, nc2;
            cos_row(a.z[i] + (long long)b * a.L, a.comb[i] + (long long)b * a.L, a.L, dot, na2, nc2);
            const float d = 1.f - dot / (sqrtf(na2) * sqrtf(nc2));
            v[i] = d * d;
        }
    }
    put_partial(a, S_AL0, v[0], red);
    put_partial(a, S_AL1, v[1], red);
}

// E = comb0 - F comb1 (only when F is given); written to ch[0] as input of the F^T E product
__global__ __launch_bounds__(256) void latent_fres_kernel(LatDev a) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e < a.B * a.L) a.ch[0][e] = a.comb[0][e] - a.fc1[e];
}

// gradient of the alignment term w.r.t. z_i (its negative is the gradient w.r.t. comb_i for the
// euclidean form; the cosine form has its own comb gradient)
__device__ __forceinline__ void align_grads(const LatDev& a, int i, int e, int b, int l, float w_al, float invBL,
                                            float& gz, float& gc) {
    const float zi = a.z[i][e], ci = a.comb[i][e];
    if (!a.cosine) {
        gz = w_al * 2.f * (zi - ci) * invBL;
        gc = -gz;
    } else {
        float dot, na2, nc2;
        cos_row(a.z[i] + (long long)b * a.L, a.comb[i] + (long long)b * a.L, a.L, dot, na2, nc2);
        const float na = sqrtf(na2), nc = sqrtf(nc2);
        const float s = dot / (na * nc);
        const float kap = -2.f * w_al * (1.f - s) * invBL;
        gz = kap * (ci / (na * nc) - s * zi / na2);
        gc = kap * (zi / (na * nc) - s * ci / nc2);
    }
}

// ---- K6: total gradient w.r.t. comb_i, H_i = G_i / den_i, dsigma and F-loss partial sums ----
__global__ __launch_bounds__(256) void latent_bwd_a_kernel(LatDev a) {
    __shared__ float red[4];
    const int B = a.B, L = a.L, n = B * L;
    const int e = blockIdx.x * 256 + threadIdx.x;
    const bool ok = e < n;
    float ds0 = 0.f, ds1 = 0.f, fsq = 0.f;
    if (ok) {
        const int b = e / L, l = e % L;
        const float invBL = 1.f / (float)n;
        const float w_al = a.hyper[2], w_f = a.hyper[3];
        const float s0 = a.sigma[0], s1 = a.sigma[1];
        const float c0 = a.comb[0][e], c1 = a.comb[1][e];
        const float E = c0 - (a.Fblk ? a.fc1[e] : 0.f);
        fsq = E * E;
        float G0 = w_f * 2.f * E * invBL;
        float G1 = a.Fblk ? -w_f * 2.f * a.fte[e] * invBL : 0.f;
        for (int t0 = 0; t0 < a.dcomb_nslab; t0 += 8) {     // (see the forward kernel)
            float u0[8], u1[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                u0[u] = t0 + u < a.dcomb_nslab ? a.dcomb[0][e + (t0 + u) * a.dcomb_slab_stride] : 0.f;
                u1[u] = t0 + u < a.dcomb_nslab ? a.dcomb[1][e + (t0 + u) * a.dcomb_slab_stride] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { G0 += u0[u]; G1 += u1[u]; }
        }
        float gz, gc;
        align_grads(a, 0, e, b, l, w_al, invBL, gz, gc);
        G0 += gc;
        align_grads(a, 1, e, b, l, w_al, invBL, gz, gc);
        G1 += gc;
        const float r = a.ident ? 1.f : a.rsum[b], q = a.ident ? 1.f : a.qsum[b];
        const float H0 = G0 / (s0 + s1 * r), H1 = G1 / (s1 + s0 * q);
        a.H[0][e] = H0;
        a.H[1][e] = H1;
        ds0 = H0 * a.z[0][e] - H0 * c0 + H1 * a.cz[1][e] - q * H1 * c1;
        ds1 = H1 * a.z[1][e] - H1 * c1 + H0 * a.cz[0][e] - r * H0 * c0;
    }
    const int slots[3] = {S_DSIG0, S_DSIG1, S_F};
    float pv[3] = {ds0, ds1, fsq};
    put_partials<3>(a, slots, pv);
}

// ---- K8: d(mu|logvar); block 0 also finalises the losses and dsigma ----
__global__ __launch_bounds__(256) void latent_bwd_b_kernel(LatDev a) {
    const int B = a.B, L = a.L, n = B * L;
    const int e = blockIdx.x * 256 + threadIdx.x;
    const float invBL = 1.f / (float)n;
    const float kl_scale = a.hyper[0];
    if (e < n) {
        const int b = e / L, l = e % L;
        const float w_al = a.hyper[2];
        const float s[2] = {a.sigma[0], a.sigma[1]};
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float gz, gc;
            align_grads(a, i, e, b, l, w_al, invBL, gz, gc);
            // external upstream gradients (autograd seam: the caller's own losses on z / mu / logvar_last)
            const float dz = s[i] * (a.H[i][e] + a.ch[i][e]) + gz + (a.dz_ext[i] ? a.dz_ext[i][e] : 0.f);
            const float lv = a.lv[i][e];
            const float dmu = dz + kl_scale * a.mu[i][e] * invBL + (a.dmu_ext[i] ? a.dmu_ext[i][e] : 0.f);
            float dlv = dz * a.eps[i][e] * 0.5f * expf(0.5f * lv);
            if (i == 1 && b < 2) dlv += kl_scale * (-0.5f) * (1.f - expf(lv)) / (float)L;
            if (i == 1 && a.dlv_ext) dlv += a.dlv_ext[e];
            a.dml[i][(long long)b * 2 * L + l] = dmu;
            a.dml[i][(long long)b * 2 * L + L + l] = dlv;
            if (a.dml_bf[i]) {
                a.dml_bf[i][(long long)b * 2 * L + l] = to_bf16(dmu);
                a.dml_bf[i][(long long)b * 2 * L + L + l] = to_bf16(dlv);
            }
            if (a.dmlT_bf[i]) {
                a.dmlT_bf[i][(long long)l * B + b] = to_bf16(dmu);
                a.dmlT_bf[i][(long long)(L + l) * B + b] = to_bf16(dlv);
            }
        }
    }
    if (blockIdx.x != 0) return;
    // ---- finalise (fixed summation order) ----
    // every slot's loads first, then ONE pair of barriers for all ten sums (block_sum_n: the same wave-then-wave-order adds
    // as block_sum): slot after slot it was ten dependent round trips + twenty barriers, 10 us for this launch
    const int nblk = (n + 255) / 256;
    constexpr int NS = S_DSIG1 + 1;
    __shared__ float redn[(256 / 64 + 1) * (NS + 1)];
    float v[NS + 1];
#pragma unroll
    for (int sl = 0; sl < NS; ++sl) v[sl] = (int)threadIdx.x < nblk ? a.partials[sl * JAMIE_MAX_PARTIALS + threadIdx.x] : 0.f;
    v[NS] = (int)threadIdx.x < a.n_rec_partials ? a.rec_partials[threadIdx.x] : 0.f;
    for (int i = threadIdx.x + 256; i < nblk; i += 256)
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) v[sl] += a.partials[sl * JAMIE_MAX_PARTIALS + i];
    for (int i = threadIdx.x + 256; i < a.n_rec_partials; i += 256) v[NS] += a.rec_partials[i];
    block_sum_n<NS + 1>(v, redn);
    if (threadIdx.x == 0) {
        const float* tot = redn;
        const float rec = redn[NS];
        const float w_rec = a.hyper[1], w_al = a.hyper[2], w_f = a.hyper[3];
        const float kl = -0.5f * (tot[S_TROW0] / (float)L - tot[S_MU2_0] * invBL)
                         - 0.5f * (tot[S_TROW1] / (float)L - tot[S_MU2_1] * invBL);
        const float l_kl = kl_scale * kl;
        const float l_rec = w_rec * rec;
        const float l_al = a.cosine ? w_al * (tot[S_AL0] + tot[S_AL1]) / (float)B * (1.f / (float)L)
                                    : w_al * (tot[S_AL0] + tot[S_AL1]) * invBL;
        const float l_f = w_f * tot[S_F] * invBL;
        const float total = l_kl + l_rec + l_al + l_f;
        a.losses[0] = l_kl; a.losses[1] = l_rec; a.losses[2] = l_al; a.losses[3] = l_f;
        a.losses[4] = total;
        a.losses[5] = fminf(a.losses[5], total);
        a.dsigma[0] = tot[S_DSIG0];
        a.dsigma[1] = tot[S_DSIG1];
    }
}

static int check_common(const jamie_latent* a) {
    JAMIE_ARG(a != nullptr, "null descriptor");
    JAMIE_ARG(a->B >= 2 && a->L >= 1, "B >= 2 (KL uses rows 0,1), L >= 1");
    JAMIE_ARG((long long)a->B * a->L <= 256LL * JAMIE_MAX_PARTIALS, "B*L too large for the partial buffer");
    JAMIE_ARG(a->sigma && a->hyper && a->partials && a->rsum && a->qsum, "null pointer");
    for (int i = 0; i < 2; ++i)
        JAMIE_ARG(a->mu[i] && a->lv[i] && a->z[i] && a->eps[i] && a->comb[i] && a->cz[i], "null state buffer");
    JAMIE_ARG(a->Fblk == nullptr || (a->fc1 && a->fte), "F given but fc1/fte scratch missing");
    return 0;
}

extern "C" int jamie_latent_fwd(const jamie_latent* a, const uint64_t* rng, void* stream) {
    int rc = check_common(a);
    if (rc) return rc;
    JAMIE_ARG(a->ml[0] && a->ml[1] && a->head_bias[0] && a->head_bias[1] && a->ml_nslab >= 1, "heads input");
    JAMIE_ARG(a->ml_nslab == 1 || a->ml_slab_stride >= (long long)a->B * 2 * a->L, "ml_slab_stride too small");
    JAMIE_ARG((a->eps_in[0] && a->eps_in[1]) || rng, "rng state required when eps is not given");
    hipStream_t st = (hipStream_t)stream;
    const LatDev d = to_dev(a);
    const int nblk = (a->B * a->L + 255) / 256;
    hipLaunchKernelGGL(latent_reparam_kernel, dim3(nblk), dim3(256), 0, st, d, rng);
    if (!d.ident) {
        MmJob j0 = {a->corr, a->z[1], a->cz[0], a->rsum, 0, 1};
        MmJob j1 = {a->corr, a->z[0], a->cz[1], a->qsum, 1, 1};
        launch_mm(st, a->B, a->L, j0, j1);
    }
    hipLaunchKernelGGL(latent_combine_kernel, dim3(nblk), dim3(256), 0, st, d);
    if (a->cosine)
        hipLaunchKernelGGL(latent_cosine_loss_kernel, dim3(nblk), dim3(256), 0, st, d);  // nblk >= ceil(B/256): every partial slot written
    if (a->Fblk) {
        MmJob f0 = {a->Fblk, a->comb[1], a->fc1, nullptr, 0, 1};
        MmJob f1 = {nullptr, nullptr, nullptr, nullptr, 0, 0};
        launch_mm(st, a->B, a->L, f0, f1);
    }
    return jamie_launch_status("jamie_latent_fwd");
}

extern "C" int jamie_latent_bwd(const jamie_latent* a, void* stream) {
    int rc = check_common(a);
    if (rc) return rc;
    JAMIE_ARG(a->dcomb[0] && a->dcomb[1] && a->dcomb_nslab >= 1, "dcomb input");
    JAMIE_ARG(a->dcomb_nslab == 1 || a->dcomb_slab_stride >= (long long)a->B * a->L, "dcomb_slab_stride too small");
    JAMIE_ARG(a->H[0] && a->H[1] && a->ch[0] && a->ch[1] && a->dml[0] && a->dml[1] && a->dsigma && a->losses,
              "null output/scratch");
    JAMIE_ARG(a->n_rec_partials == 0 || a->rec_partials, "rec_partials");
    hipStream_t st = (hipStream_t)stream;
    const LatDev d = to_dev(a);
    const int nblk = (a->B * a->L + 255) / 256;
    if (a->Fblk) {
        hipLaunchKernelGGL(latent_fres_kernel, dim3(nblk), dim3(256), 0, st, d);
        MmJob f0 = {nullptr, nullptr, nullptr, nullptr, 0, 0};
        MmJob f1 = {a->Fblk, a->ch[0], a->fte, nullptr, 1, 1};
        launch_mm(st, a->B, a->L, f0, f1);
    }
    hipLaunchKernelGGL(latent_bwd_a_kernel, dim3(nblk), dim3(256), 0, st, d);
    LatDev db = d;
    if (d.ident) {        // corr H_1 = H_1, corr^T H_0 = H_0: the last kernel reads H itself
        db.ch[0] = a->H[1]; db.ch[1] = a->H[0];
    } else {
        MmJob j0 = {a->corr, a->H[1], a->ch[0], nullptr, 0, 1};
        MmJob j1 = {a->corr, a->H[0], a->ch[1], nullptr, 1, 1};
        launch_mm(st, a->B, a->L, j0, j1);
    }
    hipLaunchKernelGGL(latent_bwd_b_kernel, dim3(nblk), dim3(256), 0, st, db);
    return jamie_launch_status("jamie_latent_bwd");
}

// =================================================================================================
// M-modality latent block (M <= JAMIE_MAX_GROUP) for fully paired cells: identity correspondence, F = 0.
// The reference is hard-wired to two modalities (`assert len(W) == 2`, jamie.py:420; `(i + 1) % 2`,
// model.py:251-256), so this is the BUILD-DEFINED generalisation proposed in SURVEY.md §8 row A14:
//     comb     = sum_j sigma_j z_j / sum_j sigma_j                     (the same for every modality)
//     KL       = sum_i -1/2 [ mean_l(1 + lv_last[i,l] - exp(lv_last[i,l])) - mean_{b,l} mu_i^2 ]   (rows i < M)
//     CosSim   = 32 * sum_i mean_b ||z_i[b] - comb[b]||^2 / L
//     F        = mean(comb^2)
// For M = 2 it coincides with the two-modality kernels above at corr = I (tested).  No oracle in the reference:
// parity for M = 3 is pinned only against the generalised CPU oracle's autograd.
// =================================================================================================
// diagnostic build only (-DJAMIE_LAT_STAMP, tools/stamp_latent.sh): thread 0 of EVERY workgroup of the fused kernels writes
// s_memrealtime (100 MHz) at its phase boundaries into a buffer of its own (forward: slots 0..7, backward: 8..15 of the
// workgroup's 16); nothing else reads it and no stamp exists in the product build
#ifdef JAMIE_LAT_STAMP
__device__ unsigned long long jamie_lat_stamps[1024 * 16];
#define LSTAMP(a, k) do { if (threadIdx.x == 0 && blockIdx.x < 1024) jamie_lat_stamps[blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" int jamie_latent_debug_stamps(unsigned long long* host_out, int n_blocks) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(jamie_lat_stamps), sizeof(unsigned long long) * 16 * n_blocks);
}
#else
#define LSTAMP(a, k) do {} while (0)
#endif
#define LF_NT 1024          // threads per workgroup: 16 waves, so that every SIMD has four waves to hide latencies behind
                            // (four waves of four elements each ran this launch in 36 us: one wave per SIMD pays every
                            // dependent VALU / LDS / memory latency in full)

struct LatMDev {
    int B, L, M;
    const float* ml[LM]; int ml_nslab; long long ml_slab_stride;
    const float* head_bias[LM]; const float* eps_in[LM];
    const float* sigma; const float* hyper;
    float* mu[LM]; float* lv[LM]; float* z[LM]; float* eps[LM]; float* comb;
    float* partials;
    const float* dcomb[LM]; int dcomb_nslab; long long dcomb_slab_stride;
    float* dml[LM]; float* dsigma;
    const float* rec_partials; int n_rec_partials; float* losses;
    int rng_stream;
    // fused tail: decoder layer 0 (g1_i = comb W_i^T + b_i), bf16 copies, head-bias gradients
    float* g1[LM]; const float* dec0_W[LM]; const float* dec0_b[LM]; int d[LM];
    float* comb_alias[LM];
    unsigned short* comb_bf16[LM]; unsigned short* combT_bf16[LM];
    unsigned short* dml_bf16[LM]; unsigned short* dmlT_bf16[LM];
    float* dbias_head[LM]; float* colpart; int accumulate;
    unsigned* ticket;             // zero-initialised counter of finished backward workgroups (reset by the last one)
    int defer_final;              // the caller finalises later (jamie_grad_sqnorm_ranges_fin): no ticket, no finalisation here
    LatFinal fin;
    int chunk_begin[LM + 1];      // column chunks of the decoder product, per modality (prefix sums)
    // fused tail of the backward launch: da2_i [B, d_i] = d(mu | logvar)_i [B, 2L] head_W_i [2L, d_i] (the heads' input gradient)
    const float* head_W[LM]; float* da2[LM];
    unsigned short* dec0_WT[LM];  // optional: bf16 [L, d] transposed copy of W_dec0 (written by the forward launch's row-block-0 chunks)
    int bchunk_begin[LM + 1];     // column chunks of that product (chunk 0: the owner workgroups), prefix sums
    int g1_panel, da2_panel;      // g1 / da2 in panels of P = JAMIE_PANEL columns ((b, c) at ((c / P) * B + b) * P + c % P): what BatchNorm reads
    // heads product inside the forward launch (round 5): mu | logvar = a2_i [B, d_i] (bf16) x head_W16_i [2L, d_i]^T (bf16), fp32
    // accumulation -- what the heads GEMM launch + its split-K slabs were (`ml` unused then)
    const unsigned short* heads_a[LM]; const unsigned short* heads_W[LM];
};

// ---- forward: ONE launch from the heads' split-K slabs to the decoder's first pre-activation ----
// Workgroup (rb, chunk): the 32 cells rb*32 .. of the batch and one chunk of COLS output columns of one modality's decoder
// layer 0.  Phase A (every workgroup, redundantly per chunk: 32 x L elements per modality, a few KB from L2): mu | logvar =
// sum of the slabs + bias, eps (given, or Philox keyed by the element: the same numbers in every workgroup), z, comb; the
// workgroups of chunk 0 store them, the bf16 copies and the loss partial sums of their 32 cells.  Phase B: g1[32, COLS] =
// comb[32, L] W[COLS, L]^T + b in exact fp32 on the vector ALU (K = L <= 128: 0.2 GFLOP per step in all, not MFMA work):
// comb and the W chunk sit in LDS (W rows padded by one float: conflict-free), a thread owns one column and RPT rows.
// Replaces four launches of the two-modality path (reparameterise, combine, bf16 casts, the [B, L] x [L, d] GEMM).
// MM = the number of modalities as a template constant: the per-modality register arrays of phase A are sized for LM = 4, and with
// a run-time M the compiler keeps all four alive (spills at 1024 threads)
template <int LMAX, int COLS, int MM>
__global__ __launch_bounds__(LF_NT) void latent_m_fwd_kernel(LatMDev a, const uint64_t* rng) {
    constexpr int EPT = LF_ROWS * LMAX / LF_NT;        // elements per thread and modality in phase A (upper bound)
    constexpr int WV = COLS * LMAX / 4 / LF_NT;        // float4 of the W chunk per thread (upper bound)
    constexpr int TRIP = LMAX * MM <= 64 ? 8 : (LMAX * MM <= 128 ? 4 : 2);      // slabs whose loads are in flight together
    __shared__ float Ws[COLS][LMAX + 1];
    __shared__ float Cs[LF_ROWS][LMAX + 2];            // (+2: the MFMA operand read Cs[lane & 31][k + (lane >> 5)] is conflict-free)
    __shared__ float red[(LF_NT / 64 + 1) * (3 * LM + 1)];
    // heads product (heads_a given): per-wave partial sums [waves / 2][col tiles][16][64] for the tree over the 16 waves' K
    // ranges, then the row block's mu | logvar of every modality [MM][LF_ROWS][2 LMAX + 1]; LMAX <= 64 only (128: the GEMM launch)
    constexpr int HT = LMAX <= 64 ? 2 * LMAX / 32 : 1;                      // 32-column tiles of mu | logvar
    __shared__ float Hp[LMAX <= 64 ? 4 * HT * 16 * 64 : 1];               // four waves' partial sums at a time
    __shared__ float Ms[LMAX <= 64 ? LF_ROWS * (2 * LMAX + 1) : 1];        // one modality's mu | logvar of the row block
    constexpr int M = MM;
    const int B = a.B, L = a.L, tid = threadIdx.x;
    const int n_rb = (B + LF_ROWS - 1) / LF_ROWS;
    const int rb = blockIdx.x % n_rb, chunk = blockIdx.x / n_rb;
    const int r0 = rb * LF_ROWS;
    int mi = 0;
#pragma unroll
    for (int i = 1; i < LM; ++i)
        if (i < M && chunk >= a.chunk_begin[i]) mi = i;
    const int c0 = (chunk - a.chunk_begin[mi]) * COLS;
    // chunk 0 is the OWNER of the 32 cells: it stores the latents, their bf16 copies and the loss partial sums and computes no
    // decoder columns (those workgroups would otherwise be the last to finish: stores + 13 block sums + the product);
    // chunks 1.. are the decoder products
    const bool owner = chunk == 0;
    const bool dec = !owner && a.g1[mi] != nullptr;
    LSTAMP(a, 0);
    // the W chunk's loads (and the bias of this lane's output column) go out first: their latency hides under phase A
    // (L % 4 != 0: a W row is not 16-byte aligned and its last quad is ragged -- element loads, the tail zero-filled)
    const int L4 = (L + 3) >> 2;
    const bool lvec = (L & 3) == 0;
    float4 wreg[WV];
    float bias = 0.f;
    const int ocol = c0 + ((tid >> 6) % (COLS / 32)) * 32 + (tid & 31);      // phase B: the column of this lane's MFMA results
    if (dec) {
#pragma unroll
        for (int t = 0; t < WV; ++t) {
            const int f = tid + LF_NT * t, c = f / L4, k4 = f % L4;
            wreg[t] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (f < COLS * L4 && c0 + c < a.d[mi]) {
                const float* wp = a.dec0_W[mi] + (long long)(c0 + c) * L + 4 * k4;
                if (lvec) wreg[t] = *reinterpret_cast<const float4*>(wp);
                else wreg[t] = make_float4(wp[0], 4 * k4 + 1 < L ? wp[1] : 0.f, 4 * k4 + 2 < L ? wp[2] : 0.f, 4 * k4 + 3 < L ? wp[3] : 0.f);
            }
        }
        if (ocol < a.d[mi]) bias = a.dec0_b[mi][ocol];
    }
    // ---- phase A ----  (all loads and the arithmetic first, results in registers; the stores follow in a second loop:
    // a store between two elements' loads would serialise their latencies, the pointers may alias for all the compiler knows)
    float p_mu2[LM] = {0.f, 0.f, 0.f, 0.f}, p_trow[LM] = {0.f, 0.f, 0.f, 0.f}, p_al[LM] = {0.f, 0.f, 0.f, 0.f}, p_f = 0.f;
    float S = 0.f, sgm[LM];
#pragma unroll
    for (int i = 0; i < LM; ++i) sgm[i] = i < M ? a.sigma[i] : 0.f;       // (first used after the slab loop: no wait in front of it)
    float v_mu[EPT][LM], v_lv[EPT][LM], v_ep[EPT][LM], v_z[EPT][LM], v_comb[EPT];
    // the slab loop is the OUTER loop: one round trip per slab for all of a thread's elements (inside the element loop
    // its runtime trip count serialised 3 x EPT x M dependent round trips: 36 us for this launch)
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int el = tid + LF_NT * j;
        const int row = el / L, l = el % L, b = r0 + row;
        const bool ok = el < LF_ROWS * L && b < B;
#pragma unroll
        for (int i = 0; i < LM; ++i) {
            v_mu[j][i] = (i < M && ok) ? a.head_bias[i][l] : 0.f;
            v_lv[j][i] = (i < M && ok) ? a.head_bias[i][L + l] : 0.f;
        }
    }
    const bool heads_here = LMAX <= 64 && a.heads_a[0] != nullptr;          // (uniform)
    if constexpr (LMAX <= 64) {
    if (heads_here) {
        // ---- the heads' product for this row block (model.py:180,185: mu | logvar = a2 W_h^T), every modality: the operands are
        // what the heads GEMM launch read (bf16 a2 as BatchNorm stored it, the bf16 weight copy), fp32 accumulation.  The 16 waves
        // split K; a wave multiplies its K range on the matrix pipe (32x32x16: M = 32 columns of mu | logvar per tile, N = the 32
        // cells) with the fragments loaded straight from global memory (W_h is 2L x d: every workgroup of the launch reads all of
        // it through L2 -- 0.4 MB at config 2, the floor of this phase: ~6 us at the 55-66 GB/s a CU takes in), then the partial
        // sums are added in a fixed tree through LDS.  Replaces the heads GEMM launch and its 8 + 8 split-K slabs.
        typedef float hf32x16 __attribute__((ext_vector_type(16)));
        typedef __bf16 hbf16x8 __attribute__((ext_vector_type(8)));
        typedef unsigned int hu32x4 __attribute__((ext_vector_type(4)));
        constexpr int NWV = LF_NT / 64;
        const int wv = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
        const int tiles = (2 * L + 31) >> 5;
#pragma unroll
        for (int i = 0; i < LM; ++i) {
            if (i >= M) continue;
            const int dd = a.d[i];
            const int nks = (dd + 15) >> 4, per = (nks + NWV - 1) / NWV;
            const int k_lo = wv * per, k_hi = min(nks, k_lo + per);
            const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.heads_a[i], 0, B * dd * 2, 0x00020000);
            const __amdgpu_buffer_rsrc_t w_rs = __builtin_amdgcn_make_buffer_rsrc((void*)a.heads_W[i], 0, 2 * L * dd * 2, 0x00020000);
            const unsigned arow = (unsigned)min(r0 + r, B - 1) * (unsigned)dd * 2u + 16u * hh;      // (cells beyond B repeat the last one: unused)
            unsigned wrow[HT];
#pragma unroll
            for (int t = 0; t < HT; ++t) wrow[t] = (32 * t + r < 2 * L) ? (unsigned)(32 * t + r) * (unsigned)dd * 2u + 16u * hh : 0xFFFFFFF0u;
            hf32x16 acc[HT];
#pragma unroll
            for (int t = 0; t < HT; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
            // (KU k-steps per trip: their loads go out together; a k-step's second half may lie beyond d -- d is a multiple of 8 -- and
            //  reads as zero: the per-lane offset is swapped for an out-of-range one, the row's neighbour must not leak in)
            constexpr int KU = LMAX <= 32 ? 2 : 1;          // k-steps whose loads go out together (registers: 128 per lane at 1024 threads)
#pragma unroll 1
            for (int ks = k_lo; ks < k_hi; ks += KU) {
                hu32x4 fa[KU], fw[KU][HT];
#pragma unroll
                for (int u = 0; u < KU; ++u) {
                    const int k0 = 16 * (ks + u);
                    const bool in = ks + u < k_hi && k0 + 8 * hh < dd;
                    fa[u] = __builtin_amdgcn_raw_buffer_load_b128(a_rs, in ? (int)(arow + 2u * (unsigned)k0) : (int)0xFFFFFFF0u, 0, 0);
#pragma unroll
                    for (int t = 0; t < HT; ++t)
                        fw[u][t] = __builtin_amdgcn_raw_buffer_load_b128(w_rs, (in && wrow[t] != 0xFFFFFFF0u) ? (int)(wrow[t] + 2u * (unsigned)k0) : (int)0xFFFFFFF0u, 0, 0);
                }
#pragma unroll
                for (int u = 0; u < KU; ++u)
#pragma unroll
                    for (int t = 0; t < HT; ++t)
                        if (t < tiles)
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(hbf16x8, fw[u][t]), __builtin_bit_cast(hbf16x8, fa[u]),
                                                                            acc[t], 0, 0, 0);
            }
            // the 16 waves' sums added in a FIXED order through a 4-slot LDS buffer: waves [lo, lo + n) store, waves [to, to + n)
            // add their partner's sum: (12..15 -> 8..11), (4..7 -> 0..3), (8..11 -> 0..3), (2, 3 -> 0, 1), (1 -> 0)
            static_assert(NWV == 16, "the wave tree below is written for 16 waves");
            // (the wave index as a SCALAR and the slot's base as one pointer: with `wv - lo` left to the compiler it folded the
            //  negative constant into every access, 32 address registers per fold, all spilled)
            const int wvs = __builtin_amdgcn_readfirstlane(wv);
            auto fold = [&](int lo, int to, int n) {
                if (wvs >= lo && wvs < lo + n) {
                    float* hp = Hp + (wvs - lo) * (HT * 16 * 64) + lane;
#pragma unroll
                    for (int t = 0; t < HT; ++t)
#pragma unroll
                        for (int e = 0; e < 16; ++e) hp[(t * 16 + e) * 64] = acc[t][e];
                }
                lds_barrier();
                if (wvs >= to && wvs < to + n) {
                    const float* hp = Hp + (wvs - to) * (HT * 16 * 64) + lane;
#pragma unroll
                    for (int t = 0; t < HT; ++t)
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[t][e] += hp[(t * 16 + e) * 64];
                }
                lds_barrier();
            };
            fold(12, 8, 4); fold(4, 0, 4); fold(8, 0, 4); fold(2, 0, 2); fold(1, 0, 1);
            // wave 0 holds the sums: D[column 32 t + (e & 3) + 8 (e >> 2) + 4 hh][cell r] -> Ms[cell][column]
            if (wvs == 0) {
#pragma unroll
                for (int t = 0; t < HT; ++t)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int c = 32 * t + (e & 3) + 8 * (e >> 2) + 4 * hh;
                        if (c < 2 * L) Ms[r * (2 * LMAX + 1) + c] = acc[t][e];
                    }
            }
            lds_barrier();
#pragma unroll
            for (int j = 0; j < EPT; ++j) {
                const int el = tid + LF_NT * j;
                const int row = el / L, l = el % L, b = r0 + row;
                if (el >= LF_ROWS * L || b >= B) continue;
                v_mu[j][i] += Ms[row * (2 * LMAX + 1) + l];
                v_lv[j][i] += Ms[row * (2 * LMAX + 1) + L + l];
            }
            lds_barrier();               // (Ms is rewritten for the next modality)
        }
    }
    }
    for (int s0 = 0; s0 < (heads_here ? 0 : a.ml_nslab); s0 += TRIP) {       // TRIP slabs' loads in flight per round trip, added in slab order
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            const int el = tid + LF_NT * j;
            const int row = el / L, l = el % L, b = r0 + row;
            const bool ok = el < LF_ROWS * L && b < B;
#pragma unroll
            for (int i = 0; i < LM; ++i) {
                if (i >= M || !ok) continue;
                float tm[TRIP], tl[TRIP];
#pragma unroll
                for (int u = 0; u < TRIP; ++u) {
                    const float* p = a.ml[i] + (s0 + u) * a.ml_slab_stride + (long long)b * 2 * L;
                    tm[u] = s0 + u < a.ml_nslab ? p[l] : 0.f;
                    tl[u] = s0 + u < a.ml_nslab ? p[L + l] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < TRIP; ++u) { v_mu[j][i] += tm[u]; v_lv[j][i] += tl[u]; }
            }
        }
    }
    LSTAMP(a, 1);
#pragma unroll
    for (int i = 0; i < LM; ++i) S += sgm[i];
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int el = tid + LF_NT * j;
        const int row = el / L, l = el % L, b = r0 + row;
        const bool ok = el < LF_ROWS * L && b < B;
        const int e = ok ? b * L + l : 0;
        float num = 0.f;
        // eps ~ N(0, 1): ONE Philox call and ONE Box-Muller pair per element and PAIR of modalities (its cosine branch is the
        // even modality's draw, its sine branch the odd one's: independent normals); every workgroup of the row block draws
        // the same numbers.  (A call per modality was half of phase A's 2 us: a wave64 instruction occupies its SIMD 4 cycles.)
        float draw[LM];
#pragma unroll
        for (int i = 0; i < LM; i += 2) {
            draw[i] = draw[i + 1] = 0.f;
            if (i >= M || !ok || (a.eps_in[i] && (i + 1 >= M || a.eps_in[i + 1]))) continue;
            Philox4 r = jamie_rand4(rng, (uint32_t)(a.rng_stream + (i >> 1)), (uint64_t)e);
            jamie_box_muller(r.v[0], r.v[1], draw[i], draw[i + 1]);
        }
#pragma unroll
        for (int i = 0; i < LM; ++i) {
            v_ep[j][i] = v_z[j][i] = 0.f;
            if (i >= M || !ok) continue;
            const float mu = v_mu[j][i], lv = v_lv[j][i];
            const float ep = a.eps_in[i] ? a.eps_in[i][e] : draw[i];
            const float z = mu + ep * (expf(0.5f * lv) + 1e-7f);
            v_mu[j][i] = mu; v_lv[j][i] = lv; v_ep[j][i] = ep; v_z[j][i] = z;
            num += sgm[i] * z;
        }
        v_comb[j] = ok ? num / S : 0.f;
    }
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int el = tid + LF_NT * j;
        if (el >= LF_ROWS * L) continue;
        const int row = el / L, l = el % L, b = r0 + row;
        const float comb = v_comb[j];
        Cs[row][l] = comb;
        if (b >= B || !owner) continue;
        const int e = b * L + l;
        a.comb[e] = comb;
#pragma unroll
        for (int i = 0; i < LM; ++i) {
            if (i >= M) continue;
            a.mu[i][e] = v_mu[j][i]; a.lv[i][e] = v_lv[j][i]; a.eps[i][e] = v_ep[j][i]; a.z[i][e] = v_z[j][i];
            p_mu2[i] += v_mu[j][i] * v_mu[j][i];
            if (i == M - 1 && b < M) p_trow[b] += 1.f + v_lv[j][i] - expf(v_lv[j][i]);
            if (a.comb_alias[i]) a.comb_alias[i][e] = comb;
            if (a.comb_bf16[i]) a.comb_bf16[i][e] = to_bf16(comb);
            if (a.combT_bf16[i]) a.combT_bf16[i][(long long)l * B + b] = to_bf16(comb);
            p_al[i] += (v_z[j][i] - comb) * (v_z[j][i] - comb);
        }
        p_f += comb * comb;
    }
    LSTAMP(a, 2);
    if (owner) {           // (uniform per workgroup)
        float pv[3 * LM + 1];
#pragma unroll
        for (int i = 0; i < LM; ++i) { pv[i] = p_mu2[i]; pv[LM + i] = p_trow[i]; pv[2 * LM + i] = p_al[i]; }
        pv[3 * LM] = p_f;
        block_sum_n<3 * LM + 1>(pv, red);
        // slots SM_MU2 + i = i, SM_TROW + i = LM + i, SM_AL + i = 2 LM + i, SM_F = 3 LM
        if (tid < 3 * LM + 1) a.partials[tid * JAMIE_MAX_PARTIALS + rb] = red[tid];
        lds_barrier();
    }
    LSTAMP(a, 3);
    if (!dec) return;
    // ---- phase B ----
#pragma unroll
    for (int t = 0; t < WV; ++t) {
        const int f = tid + LF_NT * t, c = f / L4, k4 = f % L4;
        if (f < COLS * L4) {
            Ws[c][4 * k4] = wreg[t].x; Ws[c][4 * k4 + 1] = wreg[t].y; Ws[c][4 * k4 + 2] = wreg[t].z; Ws[c][4 * k4 + 3] = wreg[t].w;
        }
    }
    lds_barrier();
    // by-product: the K-contiguous bf16 copy WT[k][c0 + c] of this chunk of W_dec0 (what the skinny d comb product of the backward
    // pass multiplies by): the chunk sits in LDS anyway; row block 0's workgroups write it, 8 consecutive c per thread
    if (rb == 0 && a.dec0_WT[mi]) {
        constexpr int SEGS = COLS / 8;                     // LMAX * SEGS == LF_NT
        static_assert(LMAX * (COLS / 8) == LF_NT, "one (k, 8-column segment) per thread");
        const int k = tid / SEGS, c8 = (tid % SEGS) * 8;
        if (k < L && c0 + c8 < a.d[mi]) {                  // (d is a multiple of 8: a segment is inside the row or outside)
            unsigned pk[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                pk[q] = (unsigned)to_bf16(Ws[c8 + 2 * q][k]) | ((unsigned)to_bf16(Ws[c8 + 2 * q + 1][k]) << 16);
            *reinterpret_cast<uint4*>(a.dec0_WT[mi] + (long long)k * a.d[mi] + c0 + c8) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        }
    }
    // exact-fp32 MFMA (32x32x2): wave w < COLS / 32 owns the 32-column tile w (waves are dealt round-robin to the SIMDs; the
    // matrix pipe's time is the same however the K = L steps are split over a SIMD's waves, so the other waves just leave).
    // One column x RPT rows per thread on the vector ALU was bound by its LDS reads: 2.3 us for this phase.
    {
        constexpr int NTILE = COLS / 32;
        const int wv = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
        if (wv >= NTILE) return;
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int g0 = 0; g0 < LMAX; g0 += 16) {     // 8 MFMAs per group: the 16 operand reads go out together, ahead of them
            if (g0 >= L) continue;                  // (uniform)
            float av[8], bv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = g0 + 2 * u + h;
                av[u] = k < L ? Cs[r][k] : 0.f;
                bv[u] = k < L ? Ws[wv * 32 + r][k] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
        }
        LSTAMP(a, 4);
        if (ocol < a.d[mi]) {
            // (panel layout: a row of a panel is JAMIE_PANEL floats, the panels follow one another B rows apart)
            float* out = a.g1[mi] + (a.g1_panel ? (long long)(ocol / JAMIE_PANEL) * B * JAMIE_PANEL + (ocol % JAMIE_PANEL) : (long long)ocol);
            const long long pitch = a.g1_panel ? JAMIE_PANEL : a.d[mi];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int b = r0 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (b < B) out[(long long)b * pitch] = acc[e] + bias;
            }
        }
    }
    LSTAMP(a, 5);
}

// ---- backward: d(mu | logvar) (+ bf16 copies), per-workgroup partial sums of d(sigma) and of the head-bias gradients ----
// `smp` (optional): ONE extra workgroup draws the NEXT step's batch (jamie_latent_m_bwd_ex): the sampler is a one-workgroup job
// whose own launch costs 5 us of every step; here, in the middle of the backward pass, it is early enough for the batch gather
// to ride in the optimiser launch.  The norm kernel has not advanced the step counter yet: smp.step_add = 1.
template <int LMAX, int COLS, int MM>
__global__ __launch_bounds__(LF_NT) void latent_m_bwd_kernel(LatMDev a, SampleArgs smp, const uint64_t* state) {
    constexpr int WB_BYTES = 2 * LMAX * COLS * 4;          // the head-weight chunk of phase B: [2L][COLS] fp32 (64 KB)
    constexpr int BIG = WB_BYTES > (int)sizeof(SampleLds) ? WB_BYTES : (int)sizeof(SampleLds);
    __shared__ __attribute__((aligned(16))) unsigned char big[BIG];      // sampler workgroup: its tables; the others: Ws
    const int n_work = (int)gridDim.x - (smp.idx ? 1 : 0);
    if (smp.idx && (int)blockIdx.x == n_work) {
        jamie_sample_block(*reinterpret_cast<SampleLds*>(big), smp.idx, smp.B, smp.N, smp.offset, smp.replace, state,
                           smp.rng_stream, smp.step_add);
        return;
    }
    constexpr int EPT = LF_ROWS * LMAX / LF_NT;
    constexpr int WV = 2 * LMAX * COLS / 4 / LF_NT;        // float4 of the head-weight chunk per thread (upper bound)
    __shared__ float red[(LF_NT / 64 + 1) * (SM_SLOTS + 2)];
    __shared__ float T[LF_ROWS][2 * LMAX + 2];           // one modality's d(mu | logvar) of this workgroup's cells (+2: the MFMA
                                                           // operand read T[lane & 31][k + (lane >> 5)] is conflict-free)
    constexpr int M = MM;
    const int B = a.B, L = a.L, n = B * L, tid = threadIdx.x;
    // workgroup (rb, chunk): chunk 0 = the OWNER of the 32 cells (stores, partial sums); chunks 1.. recompute the cells'
    // d(mu | logvar) (a few KB from L2) and multiply one modality's by COLS columns of its head weight: the heads' input
    // gradient without a GEMM launch of its own (K = 2L: 0.2 GFLOP per step, exact-fp32 MFMA)
    const int n_rb = (B + LF_ROWS - 1) / LF_ROWS;
    const int rb = blockIdx.x % n_rb, chunk = blockIdx.x / n_rb, r0 = rb * LF_ROWS;
    const bool owner = chunk == 0;
    int mi = 0;
#pragma unroll
    for (int i = 1; i < LM; ++i)
        if (i < M && chunk >= a.bchunk_begin[i]) mi = i;
    const int c0 = owner ? 0 : (chunk - a.bchunk_begin[mi]) * COLS;
    const int K2 = 2 * L, C4 = COLS / 4;
    float4 wreg[WV];
    auto load_w = [&]() {
        const int dd = a.d[mi];
#pragma unroll
        for (int t = 0; t < WV; ++t) {
            const int f = tid + LF_NT * t, k = f / C4, c4 = f % C4;
            wreg[t] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < K2 && c0 + 4 * c4 < dd)            // (d is a multiple of 4: a float4 is inside the row or outside)
                wreg[t] = *reinterpret_cast<const float4*>(a.head_W[mi] + (long long)k * dd + c0 + 4 * c4);
        }
    };
    // the weight chunk's loads go out first: their latency hides under phase A (L > 64: phase A needs the registers)
    if (!owner && LMAX < 128) load_w();
    const float invBL = 1.f / (float)n;
    const float kl_scale = a.hyper[0], w_al = a.hyper[2], w_f = a.hyper[3];
    float ds[LM] = {0.f, 0.f, 0.f, 0.f};
    LSTAMP(a, 8);
    float S = 0.f, sgm[LM];
#pragma unroll
    for (int i = 0; i < LM; ++i) sgm[i] = i < M ? a.sigma[i] : 0.f;       // (first used after the slab loop)
    // loads and arithmetic first (see the forward kernel), results in registers; the slab loop of the upstream gradient
    // d comb (every modality's decoder contributes; split-K slabs) is the outer loop: one round trip per slab
    constexpr int TRIP = LMAX * MM <= 64 ? 8 : (LMAX * MM <= 128 ? 4 : 2);      // slabs whose loads are in flight together
    constexpr bool PRE = LMAX <= 64;                     // the cells' saved state is loaded in front of the slab loop (registers permitting)
    float v_dmu[EPT][LM], v_dlv[EPT][LM], v_up[EPT];
    float s_comb[EPT], s_z[EPT][LM], s_lv[EPT][LM], s_mu[EPT][LM], s_ep[EPT][LM];
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        v_up[j] = 0.f;
        if (!PRE) continue;
        const int el = tid + LF_NT * j;
        const int row = el / L, l = el % L, b = r0 + row;
        const bool ok = el < LF_ROWS * L && b < B;
        const int e = ok ? b * L + l : 0;
        s_comb[j] = a.comb[e];
#pragma unroll
        for (int i = 0; i < LM; ++i) {
            if (i >= M) continue;
            s_z[j][i] = a.z[i][e]; s_lv[j][i] = a.lv[i][e]; s_mu[j][i] = a.mu[i][e]; s_ep[j][i] = a.eps[i][e];
        }
    }
    for (int s0 = 0; s0 < a.dcomb_nslab; s0 += TRIP) {
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            const int el = tid + LF_NT * j;
            const int row = el / L, l = el % L, b = r0 + row;
            if (el >= LF_ROWS * L || b >= B) continue;
            float tu[LM][TRIP];
#pragma unroll
            for (int i = 0; i < LM; ++i)
#pragma unroll
                for (int u = 0; u < TRIP; ++u)
                    tu[i][u] = (i < M && s0 + u < a.dcomb_nslab) ? a.dcomb[i][(long long)b * L + l + (s0 + u) * a.dcomb_slab_stride] : 0.f;
#pragma unroll
            for (int u = 0; u < TRIP; ++u)
#pragma unroll
                for (int i = 0; i < LM; ++i) v_up[j] += tu[i][u];
        }
    }
    LSTAMP(a, 10);
#pragma unroll
    for (int i = 0; i < LM; ++i) S += sgm[i];
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const int el = tid + LF_NT * j;
        const int row = el / L, l = el % L, b = r0 + row;
        const bool ok = el < LF_ROWS * L && b < B;
        const int e = ok ? b * L + l : 0;
#pragma unroll
        for (int i = 0; i < LM; ++i) v_dmu[j][i] = v_dlv[j][i] = 0.f;
        if (!ok) continue;
        const float comb = PRE ? s_comb[j] : a.comb[e];
        float G = w_f * 2.f * comb * invBL + v_up[j];         // F loss acts on combined[0]; + the decoders' d comb
        float ga[LM], zz[LM];
#pragma unroll
        for (int i = 0; i < LM; ++i) {
            if (i >= M) continue;
            zz[i] = PRE ? s_z[j][i] : a.z[i][e];
            ga[i] = w_al * 2.f * (zz[i] - comb) * invBL;      // d CosSim / d z_i ;  -ga[i] is d / d comb_i
            G -= ga[i];
        }
#pragma unroll
        for (int i = 0; i < LM; ++i) {
            if (i >= M) continue;
            const float dz = sgm[i] / S * G + ga[i];
            const float lv = PRE ? s_lv[j][i] : a.lv[i][e];
            const float dmu = dz + kl_scale * (PRE ? s_mu[j][i] : a.mu[i][e]) * invBL;
            float dlv = dz * (PRE ? s_ep[j][i] : a.eps[i][e]) * 0.5f * expf(0.5f * lv);
            if (i == M - 1 && b < M) dlv += kl_scale * (-0.5f) * (1.f - expf(lv)) / (float)L;
            v_dmu[j][i] = dmu; v_dlv[j][i] = dlv;
            ds[i] += G * (zz[i] - comb) / S;
        }
    }
    LSTAMP(a, 11);
    if (!owner) {                      // (uniform) ---- phase B: da2[32 cells, COLS] = T[32, 2L] Ws[2L, COLS] ----
        float (*Ws)[COLS] = reinterpret_cast<float (*)[COLS]>(big);
        if (LMAX >= 128) load_w();
#pragma unroll
        for (int i = 0; i < LM; ++i) {
            if (i != mi) continue;
#pragma unroll
            for (int j = 0; j < EPT; ++j) {
                const int el = tid + LF_NT * j;
                if (el >= LF_ROWS * L) continue;
                const int row = el / L, l = el % L;
                T[row][l] = v_dmu[j][i]; T[row][L + l] = v_dlv[j][i];       // (rows beyond B hold zeros)
            }
        }
#pragma unroll
        for (int t = 0; t < WV; ++t) {
            const int f = tid + LF_NT * t, k = f / C4, c4 = f % C4;
            if (k < K2) *reinterpret_cast<float4*>(&Ws[k][4 * c4]) = wreg[t];
        }
        lds_barrier();
        LSTAMP(a, 12);
        // exact-fp32 MFMA (32x32x2): wave w < COLS / 32 owns the 32-column tile w and all of K = 2L (see the forward kernel).
        // (One column x RPT rows per thread on the vector ALU was bound by its 12 LDS reads per 32 FMAs.)
        constexpr int NTILE = COLS / 32;
        const int wv = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
        if (wv >= NTILE) return;
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int g0 = 0; g0 < 2 * LMAX; g0 += 16) {     // 8 MFMAs per group: the 16 operand reads go out together, ahead of them
            if (g0 >= K2) continue;                     // (uniform)
            float av[8], bv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int k = g0 + 2 * u + h;
                av[u] = k < K2 ? T[r][k] : 0.f;
                bv[u] = k < K2 ? Ws[k][wv * 32 + r] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
        }
        LSTAMP(a, 13);
        const int col = c0 + wv * 32 + r;
        if (col < a.d[mi]) {
            float* out = a.da2[mi] + (a.da2_panel ? (long long)(col / JAMIE_PANEL) * B * JAMIE_PANEL + (col % JAMIE_PANEL) : (long long)col);
            const long long pitch = a.da2_panel ? JAMIE_PANEL : a.d[mi];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int b = r0 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (b < B) out[(long long)b * pitch] = acc[e];
            }
        }
        LSTAMP(a, 14);
        return;
    }
    // stores, and per modality the column sums over the workgroup's 32 cells through an LDS tile (rows added in order)
#pragma unroll
    for (int i = 0; i < LM; ++i) {
        if (i >= M) continue;          // uniform
#pragma unroll
        for (int j = 0; j < EPT; ++j) {
            const int el = tid + LF_NT * j;
            if (el >= LF_ROWS * L) continue;
            const int row = el / L, l = el % L, b = r0 + row;
            T[row][l] = v_dmu[j][i]; T[row][L + l] = v_dlv[j][i];       // (rows beyond B hold zeros)
            if (b >= B) continue;
            a.dml[i][(long long)b * 2 * L + l] = v_dmu[j][i];
            a.dml[i][(long long)b * 2 * L + L + l] = v_dlv[j][i];
            if (a.dml_bf16[i]) {
                a.dml_bf16[i][(long long)b * 2 * L + l] = to_bf16(v_dmu[j][i]);
                a.dml_bf16[i][(long long)b * 2 * L + L + l] = to_bf16(v_dlv[j][i]);
            }
            if (a.dmlT_bf16[i]) {
                a.dmlT_bf16[i][(long long)l * B + b] = to_bf16(v_dmu[j][i]);
                a.dmlT_bf16[i][(long long)(L + l) * B + b] = to_bf16(v_dlv[j][i]);
            }
        }
        lds_barrier();
        if (a.colpart && tid < 2 * L) {
            float v = 0.f;
#pragma unroll 8
            for (int row = 0; row < LF_ROWS; ++row) v += T[row][tid];
            a.colpart[((long long)rb * LM + i) * 2 * LMAX + tid] = v;
        }
        lds_barrier();
    }
    LSTAMP(a, 9);
    block_sum_n<LM>(ds, red);
    if (tid < M) a.partials[(SM_DSIG + tid) * JAMIE_MAX_PARTIALS + rb] = red[tid];
    LSTAMP(a, 15);
    // ---- the workgroup that finishes LAST finalises (losses, d sigma, head-bias gradients): a ticket counter instead of a
    // second launch.  Hand-off as MI355X_MICROARCH.md prescribes: every storing wave drains its stores, workgroup barrier,
    // one lane releases at agent scope and takes the ticket; the last one acquires at agent scope before anyone reads.
    if (a.defer_final) return;          // (uniform) the range-norm launch of this step finalises (jamie_grad_sqnorm_ranges_fin)
    __shared__ int is_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned ticket = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = ticket == (unsigned)(n_rb - 1);          // (only the owner workgroups take tickets)
        if (is_last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            *a.ticket = 0u;                        // for the next launch (visible at the kernel boundary)
        }
    }
    __syncthreads();
    if (is_last) latent_m_finalise(a.fin, red);
}

void jamie_latent_m_fill_final(const jamie_latent_m* a, LatFinal* f) {
    memset(f, 0, sizeof(*f));
    f->B = a->B; f->L = a->L; f->M = a->M; f->lmax = a->L <= 32 ? 32 : (a->L <= 64 ? 64 : 128);
    f->accumulate = a->accumulate; f->n_rec_partials = a->n_rec_partials;
    f->partials = a->partials; f->rec_partials = a->rec_partials; f->hyper = a->hyper; f->colpart = a->colpart;
    f->losses = a->losses; f->dsigma = a->dsigma;
    for (int i = 0; i < a->M && i < LM; ++i) f->dbias_head[i] = a->dbias_head[i];
}

static int latm_cols(int L) { return L <= 32 ? 256 : (L <= 64 ? 128 : 64); }

static int latm_to_dev(const jamie_latent_m* a, LatMDev& d) {
    JAMIE_ARG(a != nullptr, "null descriptor");
    JAMIE_ARG(a->M >= 2 && a->M <= LM, "2 <= M <= 4");
    JAMIE_ARG(a->B >= a->M && a->L >= 1 && a->L <= 128, "B >= M (KL uses rows 0..M-1), 1 <= L <= 128");
    JAMIE_ARG((a->B + LF_ROWS - 1) / LF_ROWS <= JAMIE_MAX_PARTIALS, "B too large for the partial buffer");
    JAMIE_ARG(a->sigma && a->hyper && a->partials && a->comb, "null pointer");
    memset(&d, 0, sizeof(d));
    d.B = a->B; d.L = a->L; d.M = a->M;
    const int cols = latm_cols(a->L);
    int chunks = 0;
    for (int i = 0; i < a->M; ++i) {
        JAMIE_ARG(a->mu[i] && a->lv[i] && a->z[i] && a->eps[i], "null state buffer");
        d.ml[i] = a->ml[i]; d.head_bias[i] = a->head_bias[i]; d.eps_in[i] = a->eps_in[i];
        d.mu[i] = a->mu[i]; d.lv[i] = a->lv[i]; d.z[i] = a->z[i]; d.eps[i] = a->eps[i];
        d.dcomb[i] = a->dcomb[i]; d.dml[i] = a->dml[i];
        d.g1[i] = a->g1[i]; d.dec0_W[i] = a->dec0_W[i]; d.dec0_b[i] = a->dec0_b[i]; d.d[i] = a->d[i];
        JAMIE_ARG(!a->g1[i] || (a->dec0_W[i] && a->dec0_b[i] && a->d[i] > 0 && (a->L % 4 != 0 || ((uintptr_t)a->dec0_W[i] % 16) == 0)),
                  "decoder layer 0: W (16-byte aligned when L is a multiple of 4), b, d");
        d.comb_alias[i] = a->comb_alias[i] == a->comb ? nullptr : a->comb_alias[i];
        d.dml_bf16[i] = (unsigned short*)a->dml_bf16[i]; d.dmlT_bf16[i] = (unsigned short*)a->dmlT_bf16[i];
        d.comb_bf16[i] = (unsigned short*)a->comb_bf16[i]; d.combT_bf16[i] = (unsigned short*)a->combT_bf16[i];
        d.dbias_head[i] = a->dbias_head[i];
        d.chunk_begin[i] = chunks == 0 ? 1 : chunks;          // chunk 0: the owner workgroups (no decoder columns)
        if (chunks == 0) chunks = 1;
        chunks += a->g1[i] ? (a->d[i] + cols - 1) / cols : 0;
    }
    for (int i = a->M; i <= LM; ++i) d.chunk_begin[i] = chunks;
    int bchunks = 1;
    for (int i = 0; i < a->M; ++i) {
        d.head_W[i] = a->head_W[i]; d.da2[i] = a->da2[i];
        d.dec0_WT[i] = (unsigned short*)a->dec0_WT_bf16[i];
        JAMIE_ARG(!a->dec0_WT_bf16[i] || (a->g1[i] && a->d[i] % 8 == 0 && ((uintptr_t)a->dec0_WT_bf16[i] % 16) == 0),
                  "dec0_WT_bf16: needs the fused decoder tail (g1), d a multiple of 8, 16-byte aligned");
        JAMIE_ARG(!a->da2[i] || (a->head_W[i] && a->d[i] > 0 && a->d[i] % 4 == 0 && ((uintptr_t)a->head_W[i] % 16) == 0),
                  "heads' input gradient: head_W (16-byte aligned), d a multiple of 4");
        d.bchunk_begin[i] = bchunks;
        bchunks += a->da2[i] ? (a->d[i] + cols - 1) / cols : 0;
    }
    for (int i = a->M; i <= LM; ++i) d.bchunk_begin[i] = bchunks;
    d.colpart = a->colpart; d.accumulate = a->accumulate; d.ticket = a->ticket;
    d.ml_nslab = a->ml_nslab; d.ml_slab_stride = a->ml_slab_stride;
    d.sigma = a->sigma; d.hyper = a->hyper; d.comb = a->comb; d.partials = a->partials;
    d.dcomb_nslab = a->dcomb_nslab; d.dcomb_slab_stride = a->dcomb_slab_stride;
    d.dsigma = a->dsigma; d.rec_partials = a->rec_partials; d.n_rec_partials = a->n_rec_partials;
    d.losses = a->losses; d.rng_stream = a->rng_stream;
    d.defer_final = a->defer_final;
    d.g1_panel = a->g1_panel ? 1 : 0; d.da2_panel = a->da2_panel ? 1 : 0;
    for (int i = 0; i < a->M; ++i) {
        d.heads_a[i] = (const unsigned short*)a->heads_a_bf16[i]; d.heads_W[i] = (const unsigned short*)a->heads_W_bf16[i];
        JAMIE_ARG((a->heads_a_bf16[i] != nullptr) == (a->heads_a_bf16[0] != nullptr) && (a->heads_W_bf16[i] != nullptr) == (a->heads_a_bf16[0] != nullptr),
                  "heads product: a2 and W for every modality or for none");
        JAMIE_ARG(!a->heads_a_bf16[i] || (a->L <= 64 && a->d[i] > 0 && a->d[i] % 8 == 0 && (uintptr_t)a->heads_a_bf16[i] % 16 == 0 &&
                                          (uintptr_t)a->heads_W_bf16[i] % 16 == 0 && (long long)a->B * a->d[i] * 2 < 0x7FFFFFF0LL),
                  "heads product: L <= 64, d a multiple of 8, 16-byte aligned bf16 operands");
    }
    jamie_latent_m_fill_final(a, &d.fin);
    return 0;
}

extern "C" int jamie_latent_m_fwd(const jamie_latent_m* a, const uint64_t* rng, void* stream) {
    LatMDev d;
    int rc = latm_to_dev(a, d);
    if (rc) return rc;
    bool need_rng = false;
    for (int i = 0; i < a->M; ++i) {
        JAMIE_ARG((a->ml[i] || a->heads_a_bf16[i]) && a->head_bias[i], "heads input");
        if (!a->eps_in[i]) need_rng = true;
    }
    JAMIE_ARG(a->heads_a_bf16[0] || (a->ml_nslab >= 1 && (a->ml_nslab == 1 || a->ml_slab_stride >= (long long)a->B * 2 * a->L)), "ml slabs");
    JAMIE_ARG(!need_rng || rng, "rng state required when eps is not given");
    // chunk 0 must exist and belong to a workgroup that stores the latents: with no decoder product at all there is one
    const int n_rb = (a->B + LF_ROWS - 1) / LF_ROWS;
    const int nblk = n_rb * d.chunk_begin[LM];
    hipStream_t st = (hipStream_t)stream;
#define LATM_FWD(MM)                                                                                                          \
    do {                                                                                                                      \
        if (a->L <= 32) hipLaunchKernelGGL((latent_m_fwd_kernel<32, 256, MM>), dim3(nblk), dim3(LF_NT), 0, st, d, rng);       \
        else if (a->L <= 64) hipLaunchKernelGGL((latent_m_fwd_kernel<64, 128, MM>), dim3(nblk), dim3(LF_NT), 0, st, d, rng);  \
        else hipLaunchKernelGGL((latent_m_fwd_kernel<128, 64, MM>), dim3(nblk), dim3(LF_NT), 0, st, d, rng);                  \
    } while (0)
    if (a->M == 2) LATM_FWD(2);
    else if (a->M == 3) LATM_FWD(3);
    else LATM_FWD(4);
#undef LATM_FWD
    return jamie_launch_status("jamie_latent_m_fwd");
}

static int latent_m_bwd_impl(const jamie_latent_m* a, const jamie_sample_args* smp, const uint64_t* state, void* stream) {
    LatMDev d;
    int rc = latm_to_dev(a, d);
    if (rc) return rc;
    for (int i = 0; i < a->M; ++i) JAMIE_ARG(a->dcomb[i] && a->dml[i], "dcomb / dml");
    JAMIE_ARG(a->dsigma && a->losses && a->dcomb_nslab >= 1, "null output");
    JAMIE_ARG(a->ticket != nullptr || a->defer_final, "ticket: a zero-initialised device uint32 is required");
    JAMIE_ARG(a->dcomb_nslab == 1 || a->dcomb_slab_stride >= (long long)a->B * a->L, "dcomb_slab_stride too small");
    JAMIE_ARG(a->n_rec_partials == 0 || a->rec_partials, "rec_partials");
    SampleArgs sa;
    memset(&sa, 0, sizeof(sa));
    if (smp && smp->idx) {
        JAMIE_ARG(state != nullptr, "sampler: rng state");
        JAMIE_ARG(smp->B > 0 && smp->N > 0 && (smp->replace || (smp->B <= smp->N && smp->B <= SMP_HASH / 2)),
                  "sampler: B <= N and B <= 2048 without replacement");
        JAMIE_ARG(smp->N + smp->offset <= 0x7fffffffLL, "sampler: indices must fit int32");
        sa.idx = smp->idx; sa.B = smp->B; sa.N = smp->N; sa.offset = smp->offset; sa.replace = smp->replace;
        sa.rng_stream = smp->rng_stream; sa.step_add = smp->step_add;
    }
    const int nblk = (a->B + LF_ROWS - 1) / LF_ROWS * d.bchunk_begin[LM] + (sa.idx ? 1 : 0);
    hipStream_t st = (hipStream_t)stream;
#define LATM_BWD(MM)                                                                                                               \
    do {                                                                                                                           \
        if (a->L <= 32) hipLaunchKernelGGL((latent_m_bwd_kernel<32, 256, MM>), dim3(nblk), dim3(LF_NT), 0, st, d, sa, state);       \
        else if (a->L <= 64) hipLaunchKernelGGL((latent_m_bwd_kernel<64, 128, MM>), dim3(nblk), dim3(LF_NT), 0, st, d, sa, state);  \
        else hipLaunchKernelGGL((latent_m_bwd_kernel<128, 64, MM>), dim3(nblk), dim3(LF_NT), 0, st, d, sa, state);                  \
    } while (0)
    if (a->M == 2) LATM_BWD(2);
    else if (a->M == 3) LATM_BWD(3);
    else LATM_BWD(4);
#undef LATM_BWD
    return jamie_launch_status("jamie_latent_m_bwd");
}

extern "C" int jamie_latent_m_bwd(const jamie_latent_m* a, void* stream) { return latent_m_bwd_impl(a, nullptr, nullptr, stream); }

extern "C" int jamie_latent_m_bwd_ex(const jamie_latent_m* a, const jamie_sample_args* sample, const uint64_t* state, void* stream) {
    return latent_m_bwd_impl(a, sample, state, stream);
}

/* workspace for the head-bias partial sums: floats */
extern "C" long long jamie_latent_m_colpart_size(int B, int L) {
    const int lmax = L <= 32 ? 32 : (L <= 64 ? 64 : 128);
    return (long long)((B + LF_ROWS - 1) / LF_ROWS) * LM * 2 * lmax;
}

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
        for (int s = 0; s < a.ml_nslab; ++s) {
            const float* p = a.ml[i] + s * a.ml_slab_stride + (long long)b * 2 * L;
            mu += p[l];
            lv += p[L + l];
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
    put_partial(a, S_MU2_0, mu2[0], red);
    put_partial(a, S_MU2_1, mu2[1], red);
    put_partial(a, S_TROW0, trow[0], red);
    put_partial(a, S_TROW1, trow[1], red);
}

// ---- small [B,B] x [B,L] products.  job 0: out = Mtx Z (+ row sums); job 1: out = Mtx^T Z (+ column sums).
// Mtx == nullptr means identity (out = Z, sums = 1). ----
struct MmJob { const float* Mtx; const float* Z; float* out; float* sums; int transpose; int active; };
struct MmJobs { MmJob j[2]; int B, L; };

// 256 threads = R rows x Lp latent columns (Lp = L rounded up to a power of two, R = 256 / Lp); the [B,B] matrix and Z are
// streamed through LDS in 64-column chunks (coalesced: a chunk of Z is one contiguous block), every thread adds its 64
// products in ascending column order with fmaf -- the order of the first version of this kernel (one thread per output
// walking a whole matrix row from global memory: 1024 dependent round trips, 150-290 us per launch at B = 512-1024; the
// general correspondence / F blocks of partial-correspondence training pay four to six such products per step).
#define MM_CH 64
__global__ __launch_bounds__(256) void small_mm_kernel(MmJobs js) {
    __shared__ float Msh[256 / 4][MM_CH + 1];       // R <= 64 rows (Lp >= 4)
    __shared__ float Zsh[MM_CH][64 + 1];            // Lp <= 64 per pass
    const MmJob& J = js.j[blockIdx.y];
    if (!J.active) return;
    const int B = js.B, L = js.L;
    int Lp = 4;
    while (Lp < L && Lp < 64) Lp <<= 1;
    const int R = 256 / Lp;
    const int r = threadIdx.x / Lp, lq = threadIdx.x % Lp;
    const int row0 = blockIdx.x * R, row = row0 + r;
    for (int l0 = 0; l0 < L; l0 += Lp) {            // L > 64: several passes over the latent columns
        const int l = l0 + lq;
        float acc = 0.f, sm = 0.f;
        if (J.Mtx == nullptr) {
            if (row < B && l < L) { J.out[(long long)row * L + l] = J.Z[(long long)row * L + l]; if (J.sums && l == 0) J.sums[row] = 1.f; }
            continue;
        }
        for (int c0 = 0; c0 < B; c0 += MM_CH) {
            for (int i = threadIdx.x; i < R * MM_CH; i += 256) {
                int rr, cc;
                if (!J.transpose) { rr = i / MM_CH; cc = i % MM_CH; } else { cc = i / R; rr = i % R; }
                const int gr = row0 + rr, gc = c0 + cc;
                float w = 0.f;
                if (gr < B && gc < B) w = J.transpose ? J.Mtx[(long long)gc * B + gr] : J.Mtx[(long long)gr * B + gc];
                Msh[rr][cc] = w;
            }
            for (int i = threadIdx.x; i < MM_CH * Lp; i += 256) {
                const int cc = i / Lp, ll = l0 + i % Lp, gc = c0 + cc;
                Zsh[cc][i % Lp] = (gc < B && ll < L) ? J.Z[(long long)gc * L + ll] : 0.f;
            }
            __syncthreads();
            const int nc = min(MM_CH, B - c0);
            for (int cc = 0; cc < nc; ++cc) {
                const float w = Msh[r][cc];
                acc = fmaf(w, Zsh[cc][lq], acc);
                sm += w;
            }
            __syncthreads();
        }
        if (row < B && l < L) {
            J.out[(long long)row * L + l] = acc;
            if (J.sums && l == 0) J.sums[row] = sm;
        }
    }
}

static void launch_mm(hipStream_t st, int B, int L, const MmJob& j0, const MmJob& j1) {
    MmJobs js;
    js.j[0] = j0; js.j[1] = j1; js.B = B; js.L = L;
    int Lp = 4;
    while (Lp < L && Lp < 64) Lp <<= 1;
    const int R = 256 / Lp;
    hipLaunchKernelGGL(small_mm_kernel, dim3((B + R - 1) / R, 2), dim3(256), 0, st, js);
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
        for (int s = 0; s < a.dcomb_nslab; ++s) {
            G0 += a.dcomb[0][e + s * a.dcomb_slab_stride];
            G1 += a.dcomb[1][e + s * a.dcomb_slab_stride];
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
    put_partial(a, S_DSIG0, ds0, red);
    put_partial(a, S_DSIG1, ds1, red);
    put_partial(a, S_F, fsq, red);
}

// ---- K8: d(mu|logvar); block 0 also finalises the losses and dsigma ----
__global__ __launch_bounds__(256) void latent_bwd_b_kernel(LatDev a) {
    __shared__ float red[4];
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
    const int nblk = (n + 255) / 256;
    float tot[LAT_SLOTS];
#pragma unroll
    for (int sl = 0; sl <= S_DSIG1; ++sl) {
        float v = 0.f;
        for (int i = threadIdx.x; i < nblk; i += 256) v += a.partials[sl * JAMIE_MAX_PARTIALS + i];
        tot[sl] = block_sum(v, red);
    }
    float rec = 0.f;
    for (int i = threadIdx.x; i < a.n_rec_partials; i += 256) rec += a.rec_partials[i];
    rec = block_sum(rec, red);
    if (threadIdx.x == 0) {
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
#define LM 4
enum { SM_MU2 = 0, SM_TROW = 4, SM_AL = 8, SM_F = 12, SM_DSIG = 13, SM_SLOTS = 17 };

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
};

__device__ __forceinline__ void put_partial_m(const LatMDev& a, int slot, float v, float* red) {
    const float t = block_sum(v, red);
    if (threadIdx.x == 0) a.partials[slot * JAMIE_MAX_PARTIALS + blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void latent_m_fwd_kernel(LatMDev a, const uint64_t* rng) {
    __shared__ float red[4];
    const int B = a.B, L = a.L, M = a.M, n = B * L;
    const int e = blockIdx.x * 256 + threadIdx.x;
    const bool ok = e < n;
    const int b = ok ? e / L : 0, l = ok ? e % L : 0;
    float mu2[LM] = {0.f, 0.f, 0.f, 0.f}, trow[LM] = {0.f, 0.f, 0.f, 0.f}, zz[LM] = {0.f, 0.f, 0.f, 0.f};
    float num = 0.f, S = 0.f;
#pragma unroll
    for (int i = 0; i < LM; ++i) {
        if (i >= M || !ok) continue;
        float mu = a.head_bias[i][l], lv = a.head_bias[i][L + l];
        for (int s = 0; s < a.ml_nslab; ++s) {
            const float* p = a.ml[i] + s * a.ml_slab_stride + (long long)b * 2 * L;
            mu += p[l];
            lv += p[L + l];
        }
        float ep;
        if (a.eps_in[i]) {
            ep = a.eps_in[i][e];
        } else {
            Philox4 r = jamie_rand4(rng, (uint32_t)(a.rng_stream + i), (uint64_t)e);
            float n1;
            jamie_box_muller(r.v[0], r.v[1], ep, n1);
        }
        const float z = mu + ep * (expf(0.5f * lv) + 1e-7f);
        a.mu[i][e] = mu; a.lv[i][e] = lv; a.eps[i][e] = ep; a.z[i][e] = z;
        zz[i] = z;
        mu2[i] = mu * mu;
        if (i == M - 1 && b < M) trow[b] = 1.f + lv - expf(lv);
        const float sg = a.sigma[i];
        num += sg * z;
        S += sg;
    }
    const float comb = ok ? num / S : 0.f;
    if (ok) a.comb[e] = comb;
#pragma unroll
    for (int i = 0; i < LM; ++i) {
        if (i >= M) continue;            // uniform
        put_partial_m(a, SM_MU2 + i, mu2[i], red);
        put_partial_m(a, SM_TROW + i, trow[i], red);
        put_partial_m(a, SM_AL + i, ok ? (zz[i] - comb) * (zz[i] - comb) : 0.f, red);
    }
    put_partial_m(a, SM_F, comb * comb, red);
}

__global__ __launch_bounds__(256) void latent_m_bwd_kernel(LatMDev a) {
    __shared__ float red[4];
    const int B = a.B, L = a.L, M = a.M, n = B * L;
    const int e = blockIdx.x * 256 + threadIdx.x;
    const bool ok = e < n;
    const float invBL = 1.f / (float)n;
    const float kl_scale = a.hyper[0], w_al = a.hyper[2], w_f = a.hyper[3];
    float ds[LM] = {0.f, 0.f, 0.f, 0.f};
    if (ok) {
        const int b = e / L, l = e % L;
        const float comb = a.comb[e];
        float S = 0.f, G = w_f * 2.f * comb * invBL;         // F loss acts on combined[0]
        float ga[LM], zz[LM];
#pragma unroll
        for (int i = 0; i < LM; ++i) {
            if (i >= M) continue;
            S += a.sigma[i];
            zz[i] = a.z[i][e];
            ga[i] = w_al * 2.f * (zz[i] - comb) * invBL;      // d CosSim / d z_i ;  -ga[i] is d / d comb_i
            G -= ga[i];
            for (int s = 0; s < a.dcomb_nslab; ++s) G += a.dcomb[i][e + s * a.dcomb_slab_stride];
        }
#pragma unroll
        for (int i = 0; i < LM; ++i) {
            if (i >= M) continue;
            const float dz = a.sigma[i] / S * G + ga[i];
            const float lv = a.lv[i][e];
            const float dmu = dz + kl_scale * a.mu[i][e] * invBL;
            float dlv = dz * a.eps[i][e] * 0.5f * expf(0.5f * lv);
            if (i == M - 1 && b < M) dlv += kl_scale * (-0.5f) * (1.f - expf(lv)) / (float)L;
            a.dml[i][(long long)b * 2 * L + l] = dmu;
            a.dml[i][(long long)b * 2 * L + L + l] = dlv;
            ds[i] = G * (zz[i] - comb) / S;
        }
    }
#pragma unroll
    for (int i = 0; i < LM; ++i) {
        if (i >= M) continue;
        put_partial_m(a, SM_DSIG + i, ds[i], red);
    }
    // the last block to finish cannot be known without a counter; block 0 of a SECOND tiny launch finalises
}

__global__ __launch_bounds__(256) void latent_m_final_kernel(LatMDev a) {
    __shared__ float red[4];
    const int B = a.B, L = a.L, M = a.M, n = B * L;
    const int nblk = (n + 255) / 256;
    const float invBL = 1.f / (float)n;
    float tot[SM_SLOTS];
#pragma unroll
    for (int sl = 0; sl < SM_SLOTS; ++sl) {
        float v = 0.f;
        for (int i = threadIdx.x; i < nblk; i += 256) v += a.partials[sl * JAMIE_MAX_PARTIALS + i];
        tot[sl] = block_sum(v, red);
    }
    float rec = 0.f;
    for (int i = threadIdx.x; i < a.n_rec_partials; i += 256) rec += a.rec_partials[i];
    rec = block_sum(rec, red);
    if (threadIdx.x == 0) {
        const float kl_scale = a.hyper[0], w_rec = a.hyper[1], w_al = a.hyper[2], w_f = a.hyper[3];
        float kl = 0.f, al = 0.f;
        for (int i = 0; i < M; ++i) {
            kl += -0.5f * (tot[SM_TROW + i] / (float)L - tot[SM_MU2 + i] * invBL);
            al += tot[SM_AL + i];
            a.dsigma[i] = tot[SM_DSIG + i];
        }
        const float l_kl = kl_scale * kl, l_rec = w_rec * rec, l_al = w_al * al * invBL, l_f = w_f * tot[SM_F] * invBL;
        const float total = l_kl + l_rec + l_al + l_f;
        a.losses[0] = l_kl; a.losses[1] = l_rec; a.losses[2] = l_al; a.losses[3] = l_f;
        a.losses[4] = total;
        a.losses[5] = fminf(a.losses[5], total);
    }
}

static int latm_to_dev(const jamie_latent_m* a, LatMDev& d) {
    JAMIE_ARG(a != nullptr, "null descriptor");
    JAMIE_ARG(a->M >= 2 && a->M <= LM, "2 <= M <= 4");
    JAMIE_ARG(a->B >= a->M && a->L >= 1, "B >= M (KL uses rows 0..M-1), L >= 1");
    JAMIE_ARG((long long)a->B * a->L <= 256LL * JAMIE_MAX_PARTIALS, "B*L too large for the partial buffer");
    JAMIE_ARG(a->sigma && a->hyper && a->partials && a->comb, "null pointer");
    memset(&d, 0, sizeof(d));
    d.B = a->B; d.L = a->L; d.M = a->M;
    for (int i = 0; i < a->M; ++i) {
        JAMIE_ARG(a->mu[i] && a->lv[i] && a->z[i] && a->eps[i], "null state buffer");
        d.ml[i] = a->ml[i]; d.head_bias[i] = a->head_bias[i]; d.eps_in[i] = a->eps_in[i];
        d.mu[i] = a->mu[i]; d.lv[i] = a->lv[i]; d.z[i] = a->z[i]; d.eps[i] = a->eps[i];
        d.dcomb[i] = a->dcomb[i]; d.dml[i] = a->dml[i];
    }
    d.ml_nslab = a->ml_nslab; d.ml_slab_stride = a->ml_slab_stride;
    d.sigma = a->sigma; d.hyper = a->hyper; d.comb = a->comb; d.partials = a->partials;
    d.dcomb_nslab = a->dcomb_nslab; d.dcomb_slab_stride = a->dcomb_slab_stride;
    d.dsigma = a->dsigma; d.rec_partials = a->rec_partials; d.n_rec_partials = a->n_rec_partials;
    d.losses = a->losses; d.rng_stream = a->rng_stream;
    return 0;
}

extern "C" int jamie_latent_m_fwd(const jamie_latent_m* a, const uint64_t* rng, void* stream) {
    LatMDev d;
    int rc = latm_to_dev(a, d);
    if (rc) return rc;
    bool need_rng = false;
    for (int i = 0; i < a->M; ++i) {
        JAMIE_ARG(a->ml[i] && a->head_bias[i], "heads input");
        if (!a->eps_in[i]) need_rng = true;
    }
    JAMIE_ARG(a->ml_nslab >= 1 && (a->ml_nslab == 1 || a->ml_slab_stride >= (long long)a->B * 2 * a->L), "ml slabs");
    JAMIE_ARG(!need_rng || rng, "rng state required when eps is not given");
    const int nblk = (a->B * a->L + 255) / 256;
    hipLaunchKernelGGL(latent_m_fwd_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, d, rng);
    return jamie_launch_status("jamie_latent_m_fwd");
}

extern "C" int jamie_latent_m_bwd(const jamie_latent_m* a, void* stream) {
    LatMDev d;
    int rc = latm_to_dev(a, d);
    if (rc) return rc;
    for (int i = 0; i < a->M; ++i) JAMIE_ARG(a->dcomb[i] && a->dml[i], "dcomb / dml");
    JAMIE_ARG(a->dsigma && a->losses && a->dcomb_nslab >= 1, "null output");
    JAMIE_ARG(a->dcomb_nslab == 1 || a->dcomb_slab_stride >= (long long)a->B * a->L, "dcomb_slab_stride too small");
    JAMIE_ARG(a->n_rec_partials == 0 || a->rec_partials, "rec_partials");
    const int nblk = (a->B * a->L + 255) / 256;
    hipLaunchKernelGGL(latent_m_bwd_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, d);
    hipLaunchKernelGGL(latent_m_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, d);
    return jamie_launch_status("jamie_latent_m_bwd");
}

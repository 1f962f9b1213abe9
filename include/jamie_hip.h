/* jamie_hip.h — C ABI of libjamie_hip.so: the MI355X (gfx950) kernels behind JAMIE's coupled-VAE
 * training / inference hot path.
 *
 * The reference (Oafish1/JAMIE v4.4.5) is pure Python on PyTorch ATen CPU kernels and has no FFI of its
 * own; its seam for this path is the duck-typed `model_class=` protocol (jamie/jamie.py:47,71,472-479,611)
 * plus the per-step ATen ops listed in SURVEY.md §2.1 (K1..K15).  Each entry point below replaces one
 * group of those ATen dispatches; the citation names the reference call site(s) it stands in for.
 * INTEGRATION.md shows the ctypes binding a JAMIE maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory unless marked host;
 *   - the caller (PyTorch) owns all buffers; no entry point allocates, frees or synchronises;
 *   - every launch goes to the `stream` argument (a hipStream_t passed as void*), so calls are
 *     capturable into a hipGraph;
 *   - return value: 0 on success, <0 for an argument error, >0 a hipError_t; the message is available
 *     from jamie_last_error() (thread-local);
 *   - matrices are row-major fp32 with explicit leading dimensions (elements);
 *   - `rng` = device uint64[4] {seed, step, reserved, reserved}: counter-based Philox4x32-10 streams are
 *     derived from (seed, step, stream id, element index), so forward and backward regenerate identical
 *     dropout masks without storing them.  Tests pass explicit masks / noise instead.
 */
#ifndef JAMIE_HIP_H
#define JAMIE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JAMIE_MAX_GROUP 4      /* problems per grouped launch (modalities) */
#define JAMIE_MAX_GEMM_GROUP 8 /* problems per grouped GEMM launch (modalities x {dX, dW} + a skinny layer's dW riding along) */
#define JAMIE_MAX_GEMM_GROUP_F32 12 /* ... of jamie_gemm_f32 / _cfg (fp32: the four large layers' dW of two modalities + the two skinny layers' in one launch) */

const char* jamie_last_error(void);
int jamie_version(void);
/* number of fp32 loss partial slots a GEMM epilogue / latent kernel may write (sizing helper) */
int jamie_max_partials(void);
/* number of sum-of-squares partials (dW tiles + range chunks) jamie_clip_adam* accepts */
int jamie_max_norm_partials(void);

/* ---------------------------------------------------------------------------------------------
 * GEMM (fp32 in, fp32 accumulate on v_mfma_f32_32x32x2_f32).  Replaces every nn.Linear forward
 * (`addmm`, model.py:151,161,180,185,192,197,207) and its two autograd backward products
 * (jamie.py:734).  C[M,N] = op(A)[M,K] * op(B)[K,N]:
 *   layout JAMIE_NT: A is [M,K] (lda), B is [N,K] (ldb)   -> y  = x  W^T      (forward)
 *   layout JAMIE_NN: A is [M,K] (lda), B is [K,N] (ldb)   -> dx = dy W        (input gradient)
 *   layout JAMIE_TN: A is [K,M] (lda), B is [K,N] (ldb)   -> dW = dy^T x      (weight gradient)
 * ------------------------------------------------------------------------------------------- */
enum { JAMIE_NT = 0, JAMIE_NN = 1, JAMIE_TN = 2 };

enum {
    JAMIE_EPI_STORE   = 0, /* C = acc (+ bias[n])                                  (+= if accumulate) */
    JAMIE_EPI_MSE     = 1, /* d = acc + bias[n] - X[m,n]; C = d * scale; partial[block] = sum d^2
                              (Rec loss + its gradient, jamie.py:637-641; aux0 = X, ld aux_ld)        */
    JAMIE_EPI_BN_EVAL = 2, /* eval-mode Linear+BatchNorm1d+LeakyReLU: y = (acc + bias - aux0[n]) *
                              rsqrt(aux1[n] + eps) * aux2[n] + aux3[n]; C = y > 0 ? y : slope*y
                              (aux0..3 = running_mean, running_var, gamma, beta)                       */
};

typedef struct {
    const float* A; const float* B; float* C;
    const float* bias;          /* [N] or NULL */
    const float* aux0; const float* aux1; const float* aux2; const float* aux3;
    float* partial;             /* per-block fp32 partial sums (EPI_MSE) or NULL */
    const int32_t* a_rows;      /* optional row gather for A (NT/NN: A row m -> a_rows[m]) or NULL */
    int M, N, K;
    int lda, ldb, ldc, aux_ld;
    int splitk;                 /* >= 1; slab s is written at C + s * slab_stride (no bias unless s==0) */
    long long slab_stride;
    int epi; int accumulate;
    float scale; float slope; float eps;
    float pscale;               /* EPI_MSE: partial[block] = pscale * sum d^2 (e.g. w_rec / (B*d)) */
    int b_tr;                   /* jamie_gemm_bf16 only: B is stored [K, N] row-major (N contiguous, ldb >= N) -- the dX
                                 * product dy W reads the weights W [out, in] as they are (model.py Linear backward), no
                                 * transposed copy; large-tile configurations with 128 columns (23, 24, 25) */
    int a_tr;                   /* jamie_gemm_bf16 only, with b_tr: A is stored [K, M] row-major (lda >= M) -- dW = dy^T a
                                 * reads dy [B, out] and a [B, in] as the layers produced them, no transposed activation
                                 * copies; 128 x 128 large-tile configurations (24, 25) */
    int store_nt;               /* non-temporal output stores (large-tile jamie_gemm_bf16 configurations, jamie_gemm_f32): for
                                 * outputs that are next read much later, e.g. weight gradients (read by the optimiser after the
                                 * whole backward pass) */
    int c_bf16;                 /* jamie_gemm_bf16, large-tile configurations, EPI_STORE without accumulate / split-K: C points to
                                 * bf16 [M, ldc] and the fp32 accumulators are rounded once on the way out (`partial` still sums
                                 * the squares of the fp32 values).  The weight gradients of the bf16 compute mode: 2 instead of 4
                                 * bytes per parameter written here and read again by jamie_clip_adam_g16 */
    int c_panel;                /* jamie_gemm_bf16, large-tile configurations, fp32 EPI_STORE without accumulate: every slab of C is
                                 * written in PANELS of P = JAMIE_PANEL columns -- element (m, n) at ((n / P) * M + m) * P + n % P
                                 * of its slab, slab_stride >= ceil(N / P) * P * M, ldc ignored -- the layout the BatchNorm launches
                                 * read (`panel` below): the column strip a BatchNorm workgroup owns is then whole contiguous blocks of
                                 * M x 4 P bytes per slab instead of M segments 4 N bytes apart (round 5: the pre-activations of
                                 * model.py:151-154 etc. go GEMM -> BatchNorm -> BatchNorm backward in this layout) */
} jamie_gemm_problem;
#ifndef JAMIE_PANEL
#define JAMIE_PANEL 16           /* columns per panel (16 fp32 = 64 bytes per row): a power of two >= 4.  8-column panels with 8-column BatchNorm
                                  * strips (three 256-thread workgroups per CU: 24 instead of 32 columns on the busiest CU) were built and
                                  * measured in round 5: +28 us per step, and +3 us with 8-column panels under the 16 / 32-column strips
                                  * (profiles/r05_ab_panel8_cq2_rejected.log) */
#endif
/* the panel width this library was built with (the host side lays its buffers out for it) */
int jamie_panel_width(void);

/* One launch computing up to JAMIE_MAX_GEMM_GROUP independent problems (the modalities of one layer; dX and dW together). */
int jamie_gemm_f32(const jamie_gemm_problem* problems /*host*/, int count, int layout, void* stream);
/* Same with an explicit tile configuration (tuning / benchmarks); cfg < 0 = choose by shape.  Configurations 0-19 run on the fp32
 * matrix pipe (v_mfma_f32_32x32x2_f32).  Configurations 20 (128 x 128 tiles) and 21 (256 x 128) run the SAME fp32 problem on the bf16 matrix pipe: every fp32 element is cut
 * into three bf16 pieces (x = hi + mid + lo, exact) and a product is six v_mfma_f32_32x32x16_bf16 into an fp32 accumulator -- fp32
 * inputs, fp32 outputs, error at the level of an fp32 product's own rounding (dropped terms <= 2^-21, typically 2^-24 of |a||b|); non-finite inputs
 * give NaN.  Every layout and epilogue; problems whose operands are not 16-byte aligned with leading dimensions and extents that
 * are multiples of 4 take the fp32 pipe's configuration of the same tile (17 / 15) instead.  The engine's default for the large layers
 * in fp32 mode (reference: the `addmm` / `mm` dispatches of model.py:151-207 and their autograd). */
int jamie_gemm_f32_cfg(const jamie_gemm_problem* problems /*host*/, int count, int layout, int cfg, void* stream);
/* Block tile (BM x BN) the launcher uses for a group with these maximum extents: one EPI_MSE partial is
 * written per tile, tile id = m_tile + tiles_m * n_tile. */
int jamie_gemm_tile(int layout, int max_m, int max_n, int max_k, int cfg, int* bm /*host*/, int* bn /*host*/);

/* ---------------------------------------------------------------------------------------------
 * bf16 compute mode (BASELINE config 2: bf16 compute / fp32 master weights).  Same products as jamie_gemm_f32 as
 * C[M,N] (fp32, or bf16 with c_bf16) = A[M,K] * B[N,K]^T on bf16 operands (A, B of the problem struct point to bf16;
 * lda/ldb in elements; K, lda, ldb multiples of 8; epilogues STORE and MSE), K-contiguous unless a_tr / b_tr say the operand is
 * stored k-row-major (large-tile configurations: the operands are read as their producers stored them, no transposed copies):
 *   forward A = a [B,in], B = W [out,in];  dX: A = dy [B,out], B = W [out,in] with b_tr;  dW: A = dy [B,out] with a_tr,
 *   B = a [B,in] with b_tr.  Small-tile configurations need K-contiguous (transposed) copies: jamie_cast_transpose.
 * ------------------------------------------------------------------------------------------- */
int jamie_gemm_bf16(const jamie_gemm_problem* problems /*host*/, int count, int cfg, void* stream);
int jamie_gemm_bf16_tile(int max_m, int max_n, int cfg, int* bm /*host*/, int* bn /*host*/);
typedef struct {
    const float* src;           /* [R, C] fp32, leading dimension ld, nslab slabs slab_stride apart (summed)   */
    void* dst;                  /* bf16 [R, C] (ldd) or NULL                                                    */
    void* dstT;                 /* bf16 [C, R] (ldt) or NULL                                                    */
    int R, C, ld, ldd, ldt, nslab;
    long long slab_stride;
    const void* src_bf16;       /* alternative source: bf16 [R, C] (ld), with src = NULL and nslab = 1 (a transposed  */
                                /* copy of an existing bf16 matrix, e.g. the weights the Adam kernel wrote)            */
    const int32_t* rows;        /* optional row gather: output row r reads source row rows[r] (x = data[idx],          */
                                /* jamie.py:583: the batch is gathered, cast and transposed in ONE launch)            */
    float* dst32; int ld32;     /* optional fp32 copy [R, C] of the gathered / slab-summed rows                        */
} jamie_cast_problem;
int jamie_cast_transpose(const jamie_cast_problem* problems /*host*/, int count /* <= 16 */, void* stream);

/* Reconstruction loss (jamie.py:637-641) behind a split-K x_hat GEMM: y = sum of `nslab` slabs [R, C] (contiguous,
 * bias already added by the GEMM), d = (y - x) * scale -> fp32 [R, C] and optional bf16 [R, C] / bf16 [C, R] copies;
 * partial[t] = pscale * sum (y - x)^2 over 64x64 tile t (tile index = m_tile + tiles_m * n_tile, the order of the
 * 64x64 GEMM's fused MSE epilogue), ceil(R/64) * ceil(C/64) partials per problem. */
typedef struct {
    const float* y; const float* x; float* d;   /* d: optional (NULL: only the bf16 copies / column sums leave the launch) */
    void* d_bf16; void* dT_bf16;        /* optional */
    float* partial;                     /* optional */
    int R, C, nslab; long long slab_stride;
    float scale, pscale;
    float* colpart;                     /* optional [ceil(R / 64), C]: column sums of d over each 64-row tile (rows added in order): the
                                         * decoder's output-bias gradient = column sums of d x_hat (jamie.py:734) is then a sum of
                                         * ceil(R / 64) rows instead of a second pass over the [R, C] matrix */
} jamie_mse_problem;
int jamie_mse_cast(const jamie_mse_problem* problems /*host*/, int count /* <= JAMIE_MAX_GROUP */, void* stream);

/* one column-sum problem: out[n] (+)= sum_m sum_slabs X[m, n] (jamie_colsum_group; extra workgroups of jamie_bn_act_bwd_cs and
 * jamie_grad_sqnorm_ranges_fin) */
typedef struct { const float* X; float* out; int M, N, ld, nslab; long long slab_stride; int accumulate; } jamie_colsum_problem;
/* ---------------------------------------------------------------------------------------------
 * BatchNorm1d(train) + LeakyReLU + Dropout, forward and backward, one column strip per workgroup.
 * Replaces native_batch_norm / leaky_relu / bernoulli_ + mul (model.py:152-154,162-164,193-195,198-200)
 * and their backward.  h may arrive as `nslab` split-K slabs (summed here; slab 0 receives the sum).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    float* h; int nslab; long long slab_stride;      /* pre-BN activations [B,N] (in/out: summed)   */
    const float* gamma; const float* beta;
    float* running_mean; float* running_var;         /* updated in place (momentum, unbiased var)   */
    float* save_mean; float* save_invstd;            /* [N] each, for the backward pass             */
    float* out;                                      /* [B,N] activation after dropout              */
    const uint8_t* mask;                             /* explicit 0/1 keep mask [B,N] or NULL (RNG)  */
    int B, N; int rng_stream;
    void* out_bf16;                                  /* optional bf16 [B,N] copy of `out` (out may be NULL) */
    void* outT_bf16;                                 /* optional bf16 transposed [N,B] copy (B <= 512, B % 8 == 0) */
    int panel;                                       /* 1: `h` (every slab, and the sum written back to slab 0) is in panels of
                                                      * JAMIE_PANEL columns (jamie_gemm_problem.c_panel); float4 kernels only
                                                      * (B <= 1024, N % 4 == 0); `out` / the bf16 copies / `mask` stay row-major */
} jamie_bnact_fwd_problem;

int jamie_bn_act_fwd(const jamie_bnact_fwd_problem* problems /*host*/, int count, float p_drop,
                     float momentum, float eps, float slope, const uint64_t* rng, void* stream);

/* jamie_bn_act_fwd / jamie_bn_act_bwd_cs with a PREFETCH RIDER: 64 extra workgroups (float4 kernels only; ignored otherwise) read
 * up to 8 device ranges (prefetch[i], prefetch_bytes[i]; 16-byte aligned; host arrays) with the default cache policy and discard
 * them -- what the NEXT launches stream (the weights of the next product, model.py:151 etc.; saved activations of the backward
 * pass) is then in the Infinity Cache instead of HBM-cold.  n_colsums may be 0. */
int jamie_bn_act_fwd_pf(const jamie_bnact_fwd_problem* problems /*host*/, int count, float p_drop, float momentum, float eps,
                        float slope, const uint64_t* rng, const void* const* prefetch /*host*/, const long long* prefetch_bytes /*host*/,
                        int n_prefetch, void* stream);

typedef struct {
    float* da; int nslab; long long slab_stride;     /* grad wrt activation out; dh is written to slab 0 */
    const float* h; const float* gamma; const float* beta;
    const float* save_mean; const float* save_invstd;
    float* dgamma; float* dbeta; float* dbias_lin;   /* [N] each; dbias_lin = colsum(dh) or NULL    */
    const uint8_t* mask;
    int B, N; int rng_stream; int accumulate;        /* accumulate: d{gamma,beta,bias} += */
    void* dh_bf16; void* dhT_bf16;                   /* optional bf16 [B,N] / transposed [N,B] copies of dh */
    int skip_f32;                                    /* 1: do not write the fp32 dh (bf16 copies only)     */
    int panel;                                       /* 1: `da` (every slab) and `h` are in panels of JAMIE_PANEL columns; needs
                                                      * skip_f32 (dh leaves as bf16 row-major only); float4 kernels only */
} jamie_bnact_bwd_problem;

int jamie_bn_act_bwd(const jamie_bnact_bwd_problem* problems /*host*/, int count, float p_drop,
                     float slope, const uint64_t* rng, void* stream);
/* The same launch with EXTRA workgroups that compute column sums (`colsums`: out[n] = sum_m X[m, n]; e.g. the decoder's
 * output-bias gradient, the column sums of d x_hat, jamie.py:734): a job of 47 short workgroups that would otherwise be a launch
 * of its own at the head of the backward pass runs beside this launch's long ones. */
int jamie_bn_act_bwd_cs(const jamie_bnact_bwd_problem* problems /*host*/, int count, float p_drop, float slope,
                        const uint64_t* rng, const jamie_colsum_problem* colsums /*host*/, int n_colsums, void* stream);
/* ... and with the prefetch rider of jamie_bn_act_fwd_pf (n_colsums may be 0) */
int jamie_bn_act_bwd_pf(const jamie_bnact_bwd_problem* problems /*host*/, int count, float p_drop, float slope, const uint64_t* rng,
                        const jamie_colsum_problem* colsums /*host*/, int n_colsums, const void* const* prefetch /*host*/,
                        const long long* prefetch_bytes /*host*/, int n_prefetch, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Latent block: reparameterisation (model.py:225-243), sigma-weighted combine (model.py:245-259),
 * KL / alignment ("CosSim") / F losses and their gradients (jamie.py:618-668), two modalities.
 * All [B,L] matrices are dense row-major with ld = L except `ml` / `dml` ([B,2L]: mu | logvar).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    int B, L;
    /* inputs */
    const float* ml[2]; int ml_nslab; long long ml_slab_stride;  /* heads GEMM output slabs [B,2L]   */
    const float* head_bias[2];                                   /* [2L]: b_mu | b_var                */
    const float* eps_in[2];          /* explicit N(0,1) noise [B,L] or NULL -> drawn from `rng`       */
    const float* sigma;              /* [2] */
    const float* corr;               /* [B,B] or NULL (= identity)                                    */
    const float* Fblk;               /* [B,B] row-normalised F block or NULL (= 0)                     */
    const float* hyper;              /* device float[16]: [0] kl_scale (= w_kl*32e-3*anneal), [1] w_rec, [2] w_align
                                        (= w_cos*32), [3] w_f; [8..13] = Adam block (see jamie_clip_adam)           */
    /* saved forward state (device, caller-owned) */
    float* mu[2]; float* lv[2]; float* z[2]; float* eps[2]; float* comb[2];
    float* cz[2];                    /* cz[0] = C z1, cz[1] = C^T z0                                   */
    float* rsum; float* qsum;        /* rowsum(C), colsum(C) [B]                                       */
    float* fc1;                      /* F comb1 [B,L]                                                  */
    float* partials;                 /* fp32 scratch, >= jamie_max_partials() * 16                     */
    /* backward */
    const float* dcomb[2]; int dcomb_nslab; long long dcomb_slab_stride; /* from decoder dX GEMM     */
    float* H[2]; float* ch[2];       /* scratch [B,L]: H_i = G_i / den_i;  ch[0] = C H1, ch[1] = C^T H0 */
    float* fte;                      /* scratch [B,L]: F^T E                                          */
    float* dml[2];                   /* out: d(mu|logvar) [B,2L]                                      */
    float* dsigma;                   /* out: [2]                                                       */
    const float* rec_partials; int n_rec_partials;  /* MSE partials from the decoder GEMM epilogue   */
    float* losses;                   /* out: device float[8]: KL, Rec, CosSim, F (weighted), total, running min(total) */
    int cosine;                      /* dist_method == 'cosine' (jamie.py:484-494)                    */
    int rng_stream;
    /* optional external upstream gradients [B,L] (autograd seam: the caller computes its own losses on the
       forward outputs, reference jamie.py:611-734); added to the internally derived ones; NULL = none.
       An external d(combined) is passed as one more `dcomb` slab.                                     */
    const float* dz_ext[2]; const float* dmu_ext[2]; const float* dlv_ext /* last modality's logvar */;
    /* optional bf16 copies for the bf16 compute mode (NULL = none): comb [B,L] / [L,B], d(mu|logvar) [B,2L] / [2L,B] */
    void* comb_bf16[2]; void* combT_bf16[2]; void* dml_bf16[2]; void* dmlT_bf16[2];
} jamie_latent;

int jamie_latent_fwd(const jamie_latent* a /*host*/, const uint64_t* rng, void* stream);
int jamie_latent_bwd(const jamie_latent* a /*host*/, void* stream);

/* M-modality latent block (2 <= M <= 4) for fully paired cells: identity correspondence, F = 0, euclidean alignment.
 * The reference supports two modalities only (jamie.py:420, model.py:251-256); this is the build-defined
 * generalisation of SURVEY.md §8 row A14 (comb = sum_j sigma_j z_j / sum_j sigma_j; KL rows 0..M-1 of the last
 * modality's logvar).  For M = 2 it equals jamie_latent_* at corr = NULL and is the path the training step takes for
 * identity correspondence.  L: a multiple of 4, <= 128.  Pointer arrays are indexed by modality. */
typedef struct {
    int B, L, M;
    const float* ml[4]; int ml_nslab; long long ml_slab_stride;
    const float* head_bias[4]; const float* eps_in[4];
    const float* sigma; const float* hyper;
    float* mu[4]; float* lv[4]; float* z[4]; float* eps[4];
    float* comb;                     /* [B,L]: identical for every modality                              */
    float* partials;                 /* fp32 scratch, >= jamie_max_partials() * 17                       */
    const float* dcomb[4]; int dcomb_nslab; long long dcomb_slab_stride;
    float* dml[4]; float* dsigma;    /* dsigma [M]                                                        */
    const float* rec_partials; int n_rec_partials; float* losses;
    int rng_stream;
    /* fused tail (all optional, NULL = off).  Forward: the SAME launch computes decoder layer 0 of every modality,
     * g1_i [B, d_i] = comb dec0_W_i^T + dec0_b_i (reference model.py:190; exact fp32 on the vector ALU: K = L), stores comb
     * into every `comb_alias[i]` and the bf16 copies the bf16 GEMMs of the backward pass read.  Backward: bf16 copies of
     * d(mu | logvar) and the head-bias gradients dbias_head[i] [2L] = column sums of dml_i ((+)= when `accumulate`),
     * which need `colpart` (jamie_latent_m_colpart_size() floats). */
    float* g1[4]; const float* dec0_W[4] /* [d_i, L] */; const float* dec0_b[4]; int d[4];
    float* comb_alias[4];
    void* comb_bf16[4] /* [B, L] */; void* combT_bf16[4] /* [L, B] */;
    void* dml_bf16[4] /* [B, 2L] */; void* dmlT_bf16[4] /* [2L, B] */;
    float* dbias_head[4]; float* colpart; int accumulate;
    unsigned* ticket;   /* REQUIRED for the backward launch unless defer_final: one zero-initialised device uint32 (count of
                         * finished workgroups; the last one finalises the losses, d sigma and the head-bias gradients and
                         * resets it) */
    int defer_final;    /* 1: jamie_latent_m_bwd leaves the partial sums as they are; the step's range-norm launch finalises
                         * them in an extra workgroup (jamie_grad_sqnorm_ranges_fin), off the backward pass's critical path */
    /* optional fused tail of jamie_latent_m_bwd: da2[i] [B, d[i]] (fp32, ld = d[i]) = d(mu | logvar)_i [B, 2L] head_W[i] [2L, d[i]]
     * -- the heads' input gradient (dx of nn.Linear(d, 2L), model.py:141-143) computed by extra workgroups of the same launch
     * in exact fp32; d[i] a multiple of 4.  NULL: the caller runs that product as a GEMM. */
    const float* head_W[4]; float* da2[4];
    /* optional by-product of jamie_latent_m_fwd's fused tail (g1 given): the K-contiguous bf16 copy dec0_WT_bf16[i] [L, d[i]] of
     * W_dec0 [d[i], L], written by the workgroups that stage the weight chunks anyway; what jamie_gemm_bf16_skinny multiplies
     * d g1 by for the decoder-layer-0 input gradient (d[i] a multiple of 8).  NULL: not written. */
    void* dec0_WT_bf16[4];
    int g1_panel, da2_panel;    /* 1: g1[i] / da2[i] are written in panels of JAMIE_PANEL columns (see jamie_gemm_problem.c_panel):
                                 * what the BatchNorm launches that consume them read when their `panel` is set */
    /* optional (all modalities or none; L <= 64, d[i] a multiple of 8): jamie_latent_m_fwd computes the heads' product itself --
     * mu | logvar_i = heads_a_bf16[i] [B, d[i]] (bf16: the encoder's output as its BatchNorm launch stored it) x
     * heads_W_bf16[i] [2L, d[i]]^T (bf16 copy of fc_mu | fc_var, model.py:180,185), fp32 accumulation, + head_bias -- instead of
     * summing the split-K slabs `ml` of a heads GEMM launch (ml / ml_nslab are ignored then; d[i] must be set) */
    const void* heads_a_bf16[4]; const void* heads_W_bf16[4];
} jamie_latent_m;
/* What a riding sampler draws: idx[B] = jamie_sample_indices(B, N, offset, replace, {seed, step + step_add}, rng_stream)
 * (np.random.choice of jamie/jamie.py:556).  step_add = 1 in a launch that runs before the norm kernel has advanced the step. */
typedef struct { int32_t* idx; int B; long long N; long long offset; int replace; int rng_stream; int step_add; } jamie_sample_args;
/* forward: ONE launch from the heads' split-K slabs to mu / logvar / z / comb, the loss partial sums and (fused tail) the
 * decoder's first pre-activation; backward: ONE launch (gradients + partial sums; its last workgroup finalises the losses, d sigma
 * and the head-bias gradients) */
int jamie_latent_m_fwd(const jamie_latent_m* a /*host*/, const uint64_t* rng, void* stream);
int jamie_latent_m_bwd(const jamie_latent_m* a /*host*/, void* stream);
/* jamie_latent_m_bwd with ONE extra workgroup that draws the NEXT step's batch indices (sample->step_add = 1: the norm kernel
 * has not advanced the step yet): early enough for the batch gather to ride in the optimiser launch (jamie_clip_adam_ride) */
int jamie_latent_m_bwd_ex(const jamie_latent_m* a /*host*/, const jamie_sample_args* sample /*host*/, const uint64_t* state, void* stream);
long long jamie_latent_m_colpart_size(int B, int L);

/* ---------------------------------------------------------------------------------------------
 * Optimiser: global-norm clip + Adam on one flat fp32 buffer
 * (clip_grad_norm_(params, 1) + optim.Adam.step + zero_grad, jamie.py:739-741).
 * `state` = device uint64[4] shared with `rng`: state[1] (step) is incremented by jamie_grad_sqnorm.
 * hyper (device float[16], shared with jamie_latent): [8] lr, [9] beta1, [10] beta2, [11] eps, [12] max_norm,
 * [13] grad_scale (1/world_size: the all-reduced SUM is averaged inside the update).
 * ------------------------------------------------------------------------------------------- */
int jamie_optim_blocks(long long n);   /* number of partials jamie_grad_sqnorm writes for n elements */
int jamie_grad_sqnorm(const float* g, long long n, float* partials, int n_partials, uint64_t* state,
                      void* stream);
/* The same over `count` ranges [offsets[i], offsets[i] + lengths[i]) of g only: one partial per chunk of 4096
 * elements, or of the smallest multiple of 4096 that keeps the chunk count at <= 128 (jamie_sqnorm_range_blocks() of them).  Used with the dW launches of jamie_gemm_bf16 that write
 * the sum of squares of every stored tile into `partial` (EPI_STORE + partial): together they cover the gradient, and
 * jamie_clip_adam sums all of them -- clip_grad_norm_ (jamie.py:739) without a second pass over the weight gradients. */
int jamie_grad_sqnorm_ranges(const float* g, const long long* offsets /*host*/, const long long* lengths /*host*/, int count,
                             float* partials, int n_partials, uint64_t* state, void* stream);
/* The same, and additionally g_bf16[i] = bf16(g[i]) over those ranges (same offsets): with the dW launches writing bf16
 * (jamie_gemm_problem.c_bf16) the whole gradient then exists in ONE bf16 buffer for jamie_clip_adam_g16. */
int jamie_grad_sqnorm_ranges_g16(const float* g, void* g_bf16, const long long* offsets /*host*/, const long long* lengths /*host*/,
                                 int count, float* partials, int n_partials, uint64_t* state, void* stream);
/* jamie_grad_sqnorm_ranges (g_bf16 NULL) / _g16 plus extra workgroups for work that would otherwise be launches of its own on
 * the step's critical path: ONE that finalises a deferred latent backward pass (`fin->defer_final`): losses, d sigma, head-bias
 * gradients (written into g, and g_bf16) and their sum of squares into partials[range blocks]; and, optionally, column sums
 * (`colsums`, e.g. the decoder's output-bias gradient = column sums of d x_hat, jamie.py:734), each workgroup of 64 columns
 * writing its sums into g (and g_bf16) and their squares into the next partial.  The ranges must NOT cover what the extra
 * workgroups write; n_partials = range blocks + 1 + sum(ceil(N_i / 64)). */
int jamie_grad_sqnorm_ranges_fin(const float* g, void* g_bf16, const long long* offsets /*host*/, const long long* lengths /*host*/,
                                 int count, float* partials, int n_partials, uint64_t* state,
                                 const jamie_latent_m* fin /*host*/, const jamie_colsum_problem* colsums /*host or NULL*/,
                                 int n_colsums, void* stream);
/* jamie_gemm_bf16 (tile configuration 29: the dW launches) + jamie_grad_sqnorm_ranges_fin (without column sums) in ONE launch:
 * the range chunks and the finaliser run as extra workgroups behind the GEMM tiles.  For the LAST dW launch of a backward pass:
 * every gradient the ranges cover must be complete before the launch.  `fin` may be NULL (n_partials = range blocks). */
int jamie_gemm_bf16_ranges(const jamie_gemm_problem* problems /*host*/, int count, int cfg, const float* g, void* g_bf16,
                           const long long* offsets /*host*/, const long long* lengths /*host*/, int n_ranges, float* partials,
                           int n_partials, uint64_t* state, const jamie_latent_m* fin /*host or NULL*/, void* stream);
int jamie_sqnorm_range_blocks(const long long* lengths /*host*/, int count);
int jamie_clip_adam(float* p, const float* g, float* m, float* v, long long n, const float* partials,
                    int n_partials, const float* hyper, const uint64_t* state,
                    void* p_bf16 /* optional bf16 copy of the updated parameters (same layout) or NULL */, void* stream);
/* The same two steps on a bf16 gradient: the reduced gradient of the data-parallel exchange is left in its bf16 message
 * buffer (same offsets as the flat fp32 gradient) and read from there -- no fp32 copy-back pass, 2 instead of 4 bytes of
 * gradient per parameter in both kernels. */
int jamie_grad_sqnorm_bf16(const void* g_bf16, long long n, float* partials, int n_partials, uint64_t* state, void* stream);
/* jamie_clip_adam (g fp32) / jamie_clip_adam_g16 (g_is_bf16) with EXTRA workgroups that do work of the NEXT step, which would
 * otherwise be launches of its own on the critical path, beside the 256 streaming workgroups of this ~190 us launch: either the
 * next batch's sampler (`sample`; state[1] already holds the next step's number: step_add = 0) or the next batch's row gather +
 * bf16 cast (`casts`: jamie_cast_transpose problems; the batch buffers are free once the step's last dW product has run) --
 * not both: the gather reads the sampler's output. */
int jamie_clip_adam_ride(float* p, const void* g, int g_is_bf16, float* m, float* v, long long n, const float* partials,
                         int n_partials, const float* hyper, const uint64_t* state, void* p_bf16,
                         const jamie_sample_args* sample /*host or NULL*/, const jamie_cast_problem* casts /*host or NULL*/,
                         int n_casts, void* stream);
int jamie_clip_adam_g16(float* p, const void* g_bf16, float* m, float* v, long long n, const float* partials,
                        int n_partials, const float* hyper, const uint64_t* state, void* p_bf16, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Batch assembly (jamie.py:552-604).
 * ------------------------------------------------------------------------------------------- */
/* dst[b,:] = src[idx[b],:]   (`dataset[i][random_batch[i]]`, jamie.py:583) */
int jamie_gather_rows(const float* src, long long n_rows, int d, const int32_t* idx, int B, float* dst,
                      void* stream);
/* Uniform B-subset of [0,N) without replacement (replace=0) or B draws with replacement, from `rng`
 * (device-side counterpart of np.random.choice, jamie.py:556); one workgroup, deterministic. */
int jamie_sample_indices(int32_t* idx, int B, long long N, long long offset, int replace,
                         const uint64_t* rng, int rng_stream, void* stream);
/* up to 4 independent draws in one launch (one workgroup each): the hybrid sampler's pair numbers and rows of both modalities */
int jamie_sample_indices_group(const jamie_sample_args* list /*host*/, int count, const uint64_t* rng, void* stream);
/* 'hybrid' sampler of partial-correspondence training on the device (jamie.py:559-573, corrected): slot b is, with
 * probability true_ratio, known pair pairs[pidx[b] % num_corr] = (row of modality 0, row of modality 1), else (r0[b], r1[b]);
 * pidx / r0 / r1: candidate draws from jamie_sample_indices */
int jamie_hybrid_assemble(const int32_t* pairs /*[num_corr,2]*/, const int32_t* pidx, const int32_t* r0, const int32_t* r1,
                          int B, int num_corr, float true_ratio, const uint64_t* rng, int rng_stream, int32_t* idx0,
                          int32_t* idx1, void* stream);
/* corr[a,b] = (idx0[a] == idx1[b]) row-normalised (P = I_N block, jamie.py:586-589) */
int jamie_corr_from_indices(const int32_t* idx0, const int32_t* idx1, int B, float* corr, void* stream);
/* blk[a,b] = P[idx0[a] + row_off, idx1[b] + col_off] for P in CSR form (int32 indptr / indices sorted within each
 * row, fp32 values; `indices` / `vals` may be NULL when the matrix has no entries), row-normalised when `normalise`
 * (zero rows keep divisor 1): P[idx0][:, idx1] of jamie.py:586-589 for sparse partial correspondence, no N x N array. */
int jamie_csr_block(const int32_t* indptr, const int32_t* indices, const float* vals, const int32_t* idx0,
                    const int32_t* idx1, int B0, int B1, int row_off, int col_off, int normalise, float* out /*[B0,B1]*/,
                    void* stream);
/* out[a,b] = w_blk * blk[a,b] + w_add * add[a,b] with blk = M[idx0[a] + row_off, idx1[b] + col_off] of a DENSE row-major
 * matrix (leading dimension ld), row-normalised when `normalise` (zero rows keep divisor 1); `add` may be NULL.
 * Replaces P[idx0][:, idx1] / F[idx0][:, idx1], their row normalisation and the mix corr = PF_Ratio * P + (1 - PF_Ratio) * F
 * of jamie/jamie.py:586-604 (the reference first gathers a [B, N] slab). */
int jamie_dense_block(const float* M, long long ld, const int32_t* idx0, const int32_t* idx1, int B0, int B1,
                      long long row_off, long long col_off, int normalise, float w_blk, const float* add /*[B0,B1] or NULL*/,
                      float w_add, float* out /*[B0,B1]*/, void* stream);
/* out[i] = a * x[i] + b * y[i] (y may be NULL): PF_Ratio * P + (1 - PF_Ratio) * F on [B,B] blocks (jamie/jamie.py:604) */
int jamie_axpby(float* out, float a, const float* x, float b, const float* y, long long n, void* stream);
/* out[n] (+)= sum_m X[m,n]  (bias gradients of the non-BN Linear layers) */
int jamie_colsum(const float* X, int M, int N, int ld, int nslab, long long slab_stride, float* out,
                 int accumulate, void* stream);
/* the same for up to JAMIE_MAX_GROUP matrices in one launch (the modalities) */
int jamie_colsum_group(const jamie_colsum_problem* problems /*host*/, int count, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Correspondence stage: JAMIE.Prime_Dual (jamie/jamie.py:314-414, MultiOmics branch), SURVEY.md §8(f) rank 3.
 * The seven dense products per iteration of the reference become four jamie_gemm_f32 launches issued by the
 * host (T1 = F^T (F Ky), G1 = (F Ky) T1, F Ky and G2 = Kx (F Ky) for the new F); the element-wise rest of one
 * iteration is the two entry points below.  All matrices are contiguous row-major fp32 on the device.
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    float* F;            /* [m,n] correspondence matrix, updated in place (jamie.py:339, 384) */
    const float* G1;     /* [m,n] (F Ky)(F^T F Ky) for the current F (jamie.py:358-359) */
    const float* G2;     /* [m,n] Kx F Ky for the current F (jamie.py:360) */
    float* m1;           /* [m,n] first / second moment (jamie.py:350-351, 376-377) */
    float* m2;
    float* Mu;           /* [m]  dual variable (jamie.py:343, 393) */
    float* Lambda;       /* [n]  dual variable (jamie.py:342, 394) */
    float* S;            /* [n]  slack (jamie.py:344, 386-390) */
    float* rowsum;       /* [m]  F 1   of the current F, replaced by that of the new F */
    float* colsum;       /* [n]  F^T 1 of the current F, replaced by that of the new F */
    const float* alpha;  /* [1]  scaling factor a (jamie.py:335, 397-402) */
    float* rowpart;      /* workspace, jamie_pd_workspace() elements */
    float* colpart;
    int m, n;
    float rho, epsilon;  /* UnionCom attributes (jamie.py:365, 384) */
} jamie_pd_state;
/* sizes (in floats) of the two partial-sum workspaces for an [m,n] problem */
int jamie_pd_workspace(int m, int n, long long* rowpart_elems, long long* colpart_elems);
/* One iteration's element-wise work (jamie.py:357-394 minus the products): gradient from G1, G2 and the rank-one
 * terms, moments with bias correction for `iteration` (1-based), projected step, F <- (1-eps) F + eps relu(F - step),
 * then row / column sums of the new F and the S, Mu, Lambda updates.  Deterministic (no atomics). */
int jamie_pd_step(const jamie_pd_state* state /*host*/, int iteration, void* stream);
/* alpha[0] = sum(G2 o F) * inv_trkk = tr(Kx (F Ky) F^T) / tr(Kx Kx) for G2 = Kx F Ky (jamie.py:397-402);
 * `partials`: n_partials <= 4096 floats of workspace */
int jamie_pd_alpha(const float* G2, const float* F, long long count, float* partials, int n_partials, float inv_trkk,
                   float* alpha, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Device-side preprocessing: `preclass(sample, axis=0)` of the reference (utilities.py:654-678; built at
 * jamie.py:462-465, applied at jamie.py:508), SURVEY.md §8(f) rank 4.  X is [N, d] row-major fp32 (is_f64 = 0) or
 * fp64 (is_f64 = 1) on the device.
 * ------------------------------------------------------------------------------------------------ */
/* mean[d], sd[d] (fp64; population standard deviation, numpy's two-pass order); `partials`: n_row_blocks * d doubles */
int jamie_col_stats(const void* X, int is_f64, long long N, int d, long long ld, double* partials, int n_row_blocks,
                    double* mean, double* sd, void* stream);
/* out[N,d] fp32 = (X - mean) / sd computed in fp64, NaN -> 0 (utilities.py:663-669) */
int jamie_standardise(const void* X, int is_f64, long long N, int d, long long ld, const double* mean, const double* sd,
                      float* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Data-parallel exchange: RCCL collectives over xGMI behind the C ABI (`jamie_allreduce`; 8(e): cells are
 * sharded by rows over one process per GPU and the flat gradient is summed over the ranks once per step, between
 * `batch_loss.backward()` (jamie.py:734) and `clip_grad_norm_` (jamie.py:739).  The reference has no distributed code.)
 * One foreign call per collective, recordable in a launch plan like a kernel launch.  A communicator owns an RCCL communicator
 * (ncclCommInitRank on the 128-byte id rank 0 made), a HIP stream of its own and 32 completion events ("slots"): a collective
 * is enqueued on the communicator's stream BEHIND what `stream` has launched so far, so it overlaps the kernels launched after
 * it; jamie_comm_wait(slot) makes a stream wait (on the device) for the collective last issued with that slot.  librccl is
 * bound at run time (the instance already loaded in the process, else /opt/rocm/lib): a missing library is an error from these
 * entry points, not from loading libjamie_hip.so.  dtype: 0 = fp32, 1 = bf16.  Returns 1000 + ncclResult_t for an RCCL error.
 * ------------------------------------------------------------------------------------------------ */
int jamie_comm_version(int* version /*host: RCCL's NCCL_VERSION_CODE*/);
int jamie_comm_unique_id(void* id128 /*host, 128 bytes, out*/);
int jamie_comm_create(const void* id128 /*host*/, int rank, int world, void** comm /*host, out*/);
int jamie_comm_destroy(void* comm);
/* in-place SUM of buf[0 .. count) over the ranks (the 1/world average is applied inside jamie_clip_adam: hyper[13]) */
int jamie_allreduce(void* comm, void* buf, long long count, int dtype, int slot, void* stream);
/* recv[0 .. recv_count) = SUM over ranks of send[rank * recv_count .. (rank + 1) * recv_count)  (sharded optimiser: this rank's
 * piece of a large weight region's gradient) */
int jamie_reduce_scatter(void* comm, const void* send, void* recv, long long recv_count, int dtype, int slot, void* stream);
/* recv[r * send_count .. (r + 1) * send_count) = rank r's send[0 .. send_count)  (sharded optimiser: the updated weights) */
int jamie_all_gather(void* comm, const void* send, void* recv, long long send_count, int dtype, int slot, void* stream);
int jamie_comm_wait(void* comm, int slot, void* stream);

/* ------------------------------------------------------------------------------------------------
 * EXPERIMENTS (not in libjamie_hip.so; `-DJAMIE_EXPERIMENTS`, jamie_amd.build.build_experiments() -> libjamie_hip_exp.so):
 * kernels that were built, tested against the product path and measured SLOWER inside the step.  They stay as source so
 * that the measurements in profiles/ can be repeated (tests/experiments/, tools/); nothing in jamie_amd/engine.py calls them.
 * ------------------------------------------------------------------------------------------------ */
#ifdef JAMIE_EXPERIMENTS
/* Skinny products C[M, N <= 128] (fp32, written once: no slabs) = A[M, K] B[N, K]^T, both operands K-contiguous bf16, long K:
 * the heads' forward product (fc_mus | fc_vars, model.py:180,185) and the decoder-layer-0 input gradient (autograd of
 * model.py:192) on the transposed bf16 weight copy (jamie_latent_m.dec0_WT_bf16).  One workgroup per 32 x 32 output tile, its
 * 16 waves a K slice each, MFMA fragments loaded straight from global memory, partial tiles added in wave order.  Problems:
 * EPI_STORE, splitk <= 1, no transposed-operand flags; bias optional. */
int jamie_gemm_bf16_skinny(const jamie_gemm_problem* problems /*host*/, int count, void* stream);
/* Linear forward + BatchNorm1d(train) + LeakyReLU + Dropout in ONE launch (model.py:151-154 and the three sibling blocks; bf16
 * compute mode, large-tile configurations 31 / 32): jamie_gemm_bf16 on `problems` (plain stores of `splitk` fp32 slabs, bias in
 * slab 0) whose workgroups hand their slabs over INSIDE the launch -- write-through stores, one ticket per 128-column strip -- and
 * then run jamie_bn_act_fwd's strip code on `bn[i]` (whose `h`, `nslab`, `slab_stride`, `B`, `N` must describe problem i's own
 * slab buffer; `outT_bf16` must be NULL; batch <= 512, N a multiple of 4): the same bits as the two launches.
 * mode 1: the LAST workgroup of a strip to arrive reduces the whole strip; mode 2: EVERY workgroup of a strip waits (bounded) for
 * the strip's arrivals and takes a share of its eight 16-column sub-strips by arrival order (needs the strip's tiles_m x splitk
 * workgroups co-resident: they are adjacent in dispatch order).  `tickets`: device uint32 [n_tickets >= 4 + 2 * sum_i
 * ceil(N_i / 128)], zero before the first call; the launch leaves it zero; word 0 != 0 afterwards = a bounded wait gave up. */
int jamie_gemm_bf16_bn(const jamie_gemm_problem* problems /*host*/, const jamie_bnact_fwd_problem* bn /*host*/, int count /* <= 4 */,
                       int cfg, float p_drop, float momentum, float eps, float slope, const uint64_t* rng, unsigned* tickets,
                       int n_tickets, int mode, void* stream);
/* The backward products of one Linear layer -- dX = dy W on W [out, in] as stored (b_tr) and dW = dy^T a on the activations as
 * stored (a_tr + b_tr): the autograd of model.py:151,161,192,197,207 inside jamie.py:734 -- as ONE PERSISTENT launch: `n_wg`
 * workgroups (one per CU: 4 loader waves stream operands into an LDS ring by LDS-DMA, 8 consumer waves run the MFMAs and the
 * stores; hand-off through LDS words, no barrier) each work through a static list of 128 x 128 tiles.  Same arithmetic, bit for
 * bit, as jamie_gemm_bf16 on the same problems (every problem: b_tr, EPI_STORE, no bias, no accumulate; fp32 slabs, fp32 or bf16
 * (c_bf16) results, optional per-tile sums of squares in `partial`, tile id = m_tile + tiles_m * n_tile).
 *   jamie_gemm_bf16_ring_plan: the tile lists (host): sched[w * max_items + i] = (problem << 24) | tile, -1 terminated; every
 *     problem's tiles are cut into 8 contiguous chunks (workgroups w and w + 8 share an XCD) and dealt longest-first to the least
 *     loaded workgroup of the chunk's XCD.  Depends on the problems' shapes and split-K only: computed once, kept on the device.
 *   jamie_gemm_bf16_ring: the launch; `sched` is the DEVICE copy of the plan; g .. fin as in jamie_gemm_bf16_ranges (g = NULL: no
 *     range-norm riders; with riders chunk j is taken by workgroup j before its tiles); `err`: device word, zero on entry, that
 *     a hand-off poll which gave up (bounded spins; never observed) would set. */
int jamie_gemm_bf16_ring_plan(const jamie_gemm_problem* problems /*host*/, int count, int n_wg, int max_items /* <= 48 */,
                              int32_t* sched /*host, n_wg * max_items*/);
int jamie_gemm_bf16_ring(const jamie_gemm_problem* problems /*host*/, int count, const int32_t* sched /*device*/, int n_wg,
                         int max_items, const float* g, void* g_bf16, const long long* offsets /*host*/,
                         const long long* lengths /*host*/, int n_ranges, float* partials, int n_partials, uint64_t* state,
                         const jamie_latent_m* fin /*host or NULL*/, unsigned* err, void* stream);
#endif /* JAMIE_EXPERIMENTS */

#ifdef __cplusplus
}
#endif
#endif /* JAMIE_HIP_H */

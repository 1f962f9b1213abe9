// Skinny bf16 products C[M, N] = A[M, K] B[N, K]^T with N <= 128 and a long K (both operands K-contiguous bf16, fp32 accumulate
// and output): the heads' forward product [B, d] x [2L, d]^T (reference model.py:180,185) and the decoder-layer-0 input gradient
// d comb = d g1 [B, d] x W_dec0 [d, L] (autograd of model.py:192 via jamie.py:734) on the transposed bf16 copy of W_dec0 that the
// fused latent forward launch leaves behind.  gfx950.
//
// As launches of the tiled GEMM these two were ~10 us each: 64 x 64 tiles with K cut into 8 slices to find 128 workgroups, LDS
// staging, a barrier per k-step, and 8 fp32 slabs for the consumer to sum.  Here a workgroup owns ONE 32 x 32 output tile and
// its 16 waves each take a K slice: every lane loads its MFMA fragments straight from global memory (the v_mfma_f32_32x32x16_bf16
// operand layout IS 8 consecutive k of one row: one 16-byte load per fragment; no LDS staging, no barrier in the k-loop, all of a
// slice's loads in flight together), the 16 partial tiles are added in wave order through LDS (deterministic) and the tile is
// written ONCE: no slabs.
// (experiments build only: -DJAMIE_EXPERIMENTS, jamie_amd.build.build_experiments(); not in the product library)
#ifdef JAMIE_EXPERIMENTS
#include "common.h"

typedef float sk_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 sk_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int sk_u32x4 __attribute__((ext_vector_type(4)));

struct SkinnyDev {
    const unsigned short* A; const unsigned short* B; float* C; const float* bias;
    int M, N, K, lda, ldb, ldc, tiles_n, blk_begin;
    unsigned a_bytes, b_bytes;
};
struct SkinnyGroup { SkinnyDev p[JAMIE_MAX_GEMM_GROUP]; int count; };

#define SK_NW 16
#define SK_STEPS 8                      // MFMA k-steps (16 k each) whose loads are in flight together

__global__ __launch_bounds__(SK_NW * 64) void skinny_nt_bf16_kernel(SkinnyGroup g) {
    __shared__ float red[SK_NW][16][64];                 // 64 KB: partial tiles, [wave][accumulator register][lane]
    int pi = 0;
#pragma unroll
    for (int i = 1; i < JAMIE_MAX_GEMM_GROUP; ++i)
        if (i < g.count && (int)blockIdx.x >= g.p[i].blk_begin) pi = i;
    const SkinnyDev& P = g.p[pi];
    const int t = (int)blockIdx.x - P.blk_begin;
    const int m0 = (t / P.tiles_n) * 32, n0 = (t % P.tiles_n) * 32;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, h = lane >> 5;
    // K slices of a common length (a multiple of 16) over the 16 waves
    const int kc = ((P.K + SK_NW * 16 - 1) / (SK_NW * 16)) * 16;
    const int kbeg = w * kc, kend = min(P.K, kbeg + kc);
    const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc((void*)P.A, 0, (int)P.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc((void*)P.B, 0, (int)P.b_bytes, 0x00020000);
    // rows beyond M / N read row 0 of a zero-length range: an out-of-range offset returns zeros
    const unsigned a_row = m0 + r < P.M ? (unsigned)(m0 + r) * (unsigned)P.lda * 2u : 0xFFFFFFF0u;
    const unsigned b_row = n0 + r < P.N ? (unsigned)(n0 + r) * (unsigned)P.ldb * 2u : 0xFFFFFFF0u;
    sk_f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int k0 = kbeg; k0 < kend; k0 += 16 * SK_STEPS) {
        sk_u32x4 av[SK_STEPS], bv[SK_STEPS];
#pragma unroll
        for (int s = 0; s < SK_STEPS; ++s) {             // lane (r, h) of step s: k = k0 + 16 s + 8 h .. + 7 (K is a multiple of 8)
            const int k = k0 + 16 * s + 8 * h;
            const bool ok = k < kend;
            av[s] = __builtin_amdgcn_raw_buffer_load_b128(a_rs, (ok && a_row != 0xFFFFFFF0u) ? (int)(a_row + (unsigned)k * 2u) : (int)0xFFFFFFF0u, 0, 0);
            bv[s] = __builtin_amdgcn_raw_buffer_load_b128(b_rs, (ok && b_row != 0xFFFFFFF0u) ? (int)(b_row + (unsigned)k * 2u) : (int)0xFFFFFFF0u, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);               // all of the slice's loads go out before the first MFMA waits for one
#pragma unroll
        for (int s = 0; s < SK_STEPS; ++s)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(sk_bf16x8, av[s]), __builtin_bit_cast(sk_bf16x8, bv[s]), acc, 0, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) red[w][e][lane] = acc[e];
    lds_barrier();
    // thread -> output element (row = tid / 32, col = tid % 32); C/D map: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
    const int row = tid >> 5, col = tid & 31;
    const int e = (row & 3) + 4 * (row >> 3), ln = col + 32 * ((row >> 2) & 1);
    float s = 0.f;
#pragma unroll
    for (int ww = 0; ww < SK_NW; ++ww) s += red[ww][e][ln];
    const int m = m0 + row, n = n0 + col;
    if (m < P.M && n < P.N) P.C[(long long)m * P.ldc + n] = s + (P.bias ? P.bias[n] : 0.f);
}

extern "C" int jamie_gemm_bf16_skinny(const jamie_gemm_problem* pr, int count, void* stream) {
    JAMIE_ARG(pr != nullptr && count >= 1 && count <= JAMIE_MAX_GEMM_GROUP, "1 <= count <= JAMIE_MAX_GEMM_GROUP");
    SkinnyGroup g;
    memset(&g, 0, sizeof(g));
    g.count = count;
    int blocks = 0;
    for (int i = 0; i < count; ++i) {
        const jamie_gemm_problem& s = pr[i];
        JAMIE_ARG(s.A && s.B && s.C, "null operand");
        JAMIE_ARG(s.M > 0 && s.N > 0 && s.N <= 128 && s.K > 0, "1 <= N <= 128");
        JAMIE_ARG(s.K % 8 == 0 && s.lda % 8 == 0 && s.ldb % 8 == 0 && s.lda >= s.K && s.ldb >= s.K && s.ldc >= s.N,
                  "bf16 operands K-contiguous, K / lda / ldb multiples of 8");
        JAMIE_ARG(((uintptr_t)s.A % 16) == 0 && ((uintptr_t)s.B % 16) == 0, "bf16 operands must be 16-byte aligned");
        JAMIE_ARG(s.epi == JAMIE_EPI_STORE && !s.accumulate && s.splitk <= 1 && !s.a_tr && !s.b_tr && !s.c_bf16 && !s.partial && !s.a_rows,
                  "a plain fp32 store of the whole product (no slabs, no transposed operands)");
        JAMIE_ARG(((long long)(s.M - 1) * s.lda + s.K) * 2 < 0xFFFFFFF0LL && ((long long)(s.N - 1) * s.ldb + s.K) * 2 < 0xFFFFFFF0LL,
                  "operands must stay below 4 GiB");
        SkinnyDev& d = g.p[i];
        d.A = (const unsigned short*)s.A; d.B = (const unsigned short*)s.B; d.C = s.C; d.bias = s.bias;
        d.M = s.M; d.N = s.N; d.K = s.K; d.lda = s.lda; d.ldb = s.ldb; d.ldc = s.ldc;
        d.tiles_n = (s.N + 31) / 32;
        d.blk_begin = blocks;
        blocks += ((s.M + 31) / 32) * d.tiles_n;
        d.a_bytes = (unsigned)(((long long)(s.M - 1) * s.lda + s.K) * 2);
        d.b_bytes = (unsigned)(((long long)(s.N - 1) * s.ldb + s.K) * 2);
    }
    hipLaunchKernelGGL(skinny_nt_bf16_kernel, dim3(blocks), dim3(SK_NW * 64), 0, (hipStream_t)stream, g);
    return jamie_launch_status("jamie_gemm_bf16_skinny");
}

#endif  // JAMIE_EXPERIMENTS

// bf16 GEMM on v_mfma_f32_32x32x16_bf16 with fp32 accumulation/output for JAMIE's Linear layers, gfx950.
//
// bf16 compute mode of the training step (BASELINE config 2: "bf16 compute / fp32 master"): the same
// products as gemm_f32.hip (reference model.py:151,161,180,185,192,197,207; jamie.py:734).  Every operand exists ONCE, bf16,
// row-major as its producer stored it (activations / gradients [B, features], weights [out, in]); the three products of a
// Linear layer differ in which operand is k-row-major:
//     forward   y  = a   W^T      A = a  [B, in]  (K contiguous)     B = W [out, in] (K contiguous)
//     dX        dx = dy  W        A = dy [B, out] (K contiguous)     B = W [out, in] = [K, N] as stored        (b_tr)
//     dW        dW = dy^T a       A = dy [B, out] = [K, M] as stored  B = a [B, in]  = [K, N] as stored  (a_tr + b_tr)
// A K-contiguous operand is staged as [rows][64 k] (one ds_read_b128 per fragment), a k-row-major one as [64 k][128] read
// with ds_read_b64_tr_b16 (gemm_bf16_dma2_body); only launches that fall back to the small-tile kernels below (skinny or
// ragged problems of small models) still read transposed copies made by the cast / BatchNorm kernels.
//
// Three kernels: gemm_bf16_kernel (register-staged, padded LDS rows: fallback), gemm_bf16_dma_kernel (LDS-DMA, 64x64 tiles)
// and gemm_bf16_dma2_kernel (LDS-DMA, 256x128 / 128x128 tiles, hand-counted vmcnt; optional riders; in the experiments build
// (-DJAMIE_EXPERIMENTS) also the in-launch split-K reduction + BatchNorm forward, jamie_gemm_bf16_bn: measured slower, round 3).  Grouped launch, XCD mapping, split-K slabs as in gemm_f32.hip.
#include "common.h"
#include "range_norm.h"
#ifdef JAMIE_EXPERIMENTS
#include "bn_fwd_strip.h"
#endif
#include <type_traits>

// diagnostic ablations (timing only, wrong results): 1 = no global loads in the k-loop, 2 = no MFMAs,
// 3 = no LDS writes, 4 = no fragment reads (LDS read traffic removed)
#ifndef JAMIE_GEMMB_ABL
#define JAMIE_GEMMB_ABL 0
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

struct GemmBDev {
    const unsigned short* A; const unsigned short* B; float* C; const float* bias;
    const float* aux0; float* partial;
    long long slab_stride;
    int M, N, K, lda, ldb, ldc, aux_ld;
    int splitk, kchunk, tiles_m, tiles_n, n_tiles;
    int epi, accumulate, vec;
    int b_tr;            // B is [K, N] row-major (N contiguous): staged as [k][n] rows, fragments by ds_read_b64_tr_b16
    int a_tr;            // A is [K, M] row-major (M contiguous), likewise (dW = dy^T a reads dy [B, out] and a [B, in])
    int store_nt;        // non-temporal output stores (weight gradients: next read by the optimiser, a whole backward pass later)
    int c_bf16;          // C is bf16 [M, ldc]: the fp32 accumulators are rounded once on the way out (weight gradients)
    int c_panel;         // fp32 C in panels of P = JAMIE_PANEL columns: (m, n) at ((n / P) * M + m) * P + n % P (what the BatchNorm launches read)
    unsigned a_bytes, b_bytes;
    float scale, pscale;
};
struct GemmBGroup { int ntiles[JAMIE_MAX_GEMM_GROUP]; GemmBDev p[JAMIE_MAX_GEMM_GROUP]; int count; };      // ntiles[i]: tile count of problem i (0: unused)

#ifdef JAMIE_EXPERIMENTS
// In-launch split-K reduction + BatchNorm forward (jamie_gemm_bf16_bn): per problem the BatchNorm strip descriptor whose `h` is
// the GEMM's slab buffer, and the hand-off state: tickets[0..3] = error block (word 0: a bounded wait gave up), then TWO words
// per column strip of every problem (arrivals, departures), zero at allocation and zero again when a launch ends.
#define JB_FUSE_MAX 4
struct BnFuse {
    BnFwdDev p[JB_FUSE_MAX];
    int ticket_base[JB_FUSE_MAX];
    unsigned* tickets;
    const uint64_t* rng;
    float p_drop, momentum, eps, slope;
    int mode;                 // 1: the LAST workgroup of a strip to arrive reduces the strip; 2: EVERY workgroup waits for its
                              // strip's arrivals and takes a share of the 16-column sub-strips (by arrival order)
};
#else
struct BnFuse;
#endif

#define JB_OOB 0xFFFFFFF0u
__device__ __attribute__((aligned(16))) unsigned int jb_zero16[4];      // what the lanes beyond K of a ragged k-tile's LDS-DMA read

// Transposed LDS read (ds_read_b64_tr_b16) as INLINE ASM.  The intrinsic (__builtin_amdgcn_ds_read_tr16_b64) carries no memory
// operand, so hipcc's wait-count pass assumes it may read what a pending LDS-DMA writes and puts `s_waitcnt vmcnt(0)` in front of
// every one of them: in the k-row-major products (dX, dW) the k-step's just-issued DMA was drained before the first fragment read
// -- no prefetch at any ring depth, 1.0-1.2 us per k-step against 0.64 us for the same tile with plain ds_read_b128 fragments
// (profiles/r04_stamps_probe*.log; the .s of round 3's kernel shows the wait right behind the four global_load_lds).  The asm is
// invisible to that pass; the reads are ordered by the hand-written `s_waitcnt lgkmcnt(0)` in front of the MFMAs that use them
// (jb_lds_wait, followed by a sched_barrier: cdna_hip_programming.md 5.4 rule 18).
// What the asm form owes the compiler in return (cdna_hip_programming.md 5.7 item 1, form (ii)): hipcc holds an asm
// read's destination to be written when the statement ends, so NOTHING may touch a destination between the read and the wait
// -- not the 64 -> 128-bit concatenation into an MFMA operand either (a v_mov the register allocator might emit for it would copy
// stale registers: no hardware interlock).  The two 64-bit halves of a fragment therefore stay separate values until the wait:
// jb_lds_wait_for() is the `s_waitcnt lgkmcnt(0)` and names every half it retires as a read-write operand ("+v"), so each
// consumer -- the concatenation, then the MFMA -- is data-dependent on the wait statement.  tests/test_isa_hazards.py
// disassembles the product object and fails if any instruction reads or writes a destination of a transposed read before
// the wait that follows it.
typedef short jb_s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ jb_s16x4 jb_ds_read_tr16(const unsigned char* p) {
    typedef const unsigned char __attribute__((address_space(3)))* lp_t;
    const unsigned addr = (unsigned)reinterpret_cast<size_t>((lp_t)p);
    jb_s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ void jb_lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// the wait + "these registers are only valid from here on" (asm volatile statements keep their order among themselves)
__device__ __forceinline__ void jb_lds_wait_for(jb_s16x4& a) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a) : : "memory"); }
__device__ __forceinline__ void jb_after_wait(jb_s16x4& a) { asm volatile("" : "+v"(a)); }
// Measured and rejected in round 3 (the A/B builds are gone; logs in profiles/): non-temporal LDS-DMA of the weights operand
// (r03_ab_nt_weights_rejected.log: forward launches 7 % faster, the step 18-45 us slower: the backward pass finds the weights in
// the Infinity Cache), issue priority for the long tiles of a grouped launch (r03_ab_long_tile_priority_rejected.log), write-through
// split-K slab stores (r03_ab_slab_sc1_rejected.log).

// Diagnostic build only (-DJAMIE_GEMMB_STAMP, tools/stamp_gemm_bf16.sh): thread 0 of every workgroup of the large-tile
// kernel writes s_memrealtime (100 MHz) at entry / tile 0 published / k-loop done / stores issued into a buffer of its
// own that nothing else reads.  No stamp exists in the product build.
#ifdef JAMIE_GEMMB_STAMP
#define JB_NSTAMP 8
__device__ unsigned long long jamie_dbg_stamps[8192 * JB_NSTAMP];
#define JB_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 8192) jamie_dbg_stamps[blockIdx.x * JB_NSTAMP + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define JB_STAMPV(k, v) do { if (threadIdx.x == 0 && blockIdx.x < 8192) jamie_dbg_stamps[blockIdx.x * JB_NSTAMP + (k)] = (unsigned long long)(v); } while (0)
extern "C" int jamie_debug_stamps(unsigned long long* host_out, int n_blocks) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(jamie_dbg_stamps), sizeof(unsigned long long) * JB_NSTAMP * n_blocks);
}
#else
#define JB_STAMP(k) do {} while (0)
#define JB_STAMPV(k, v) do {} while (0)
#endif

// D = register prefetch depth: the global loads of tiles kt+1 .. kt+D are in flight while tile kt is multiplied.  With
// a 64x64x64 tile a k-step is only 128 MFMA cycles per wave, far less than one HBM round trip, so D = 1 pays one
// memory latency per k-step; D = 3 divides that by three (counted vmcnt waits come from the compiler: the loads are
// builtins in program order).
template <int BM, int BN, int BK, int WM, int WN, int TAG, int D>
__global__ __launch_bounds__(WM * WN * 64) void gemm_bf16_kernel(GemmBGroup g) {
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int LDB = BK * 2 + 16;                 // LDS row stride in bytes
    constexpr int CH = BK / 8;                       // 16-byte chunks per row
    constexpr int A_SZ = BM * LDB, B_SZ = BN * LDB;  // bytes
    constexpr int LA = BM * CH / NT, LB = BN * CH / NT;
    static_assert(LA >= 1 && LB >= 1 && (BM * CH) % NT == 0 && (BN * CH) % NT == 0 && BK % 16 == 0, "tile/thread mismatch");
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * (A_SZ + B_SZ)];
    __shared__ float red[WM * WN];

    // ---- block -> (problem, tile): per-problem XCD chunks (see gemm_f32.hip) ----
    const int bid = blockIdx.x;
    const int xcd = bid & 7;
    int slot = bid >> 3;
    int pi = 0, t = 0, rot = 0;
    // (branch-free, every problem's tile count loaded up front -- unused problems hold 0; as a loop of guarded iterations this was one
    //  dependent scalar-memory round trip and three branches per problem in front of the first load: gemm_f32.hip)
    int ntl[JAMIE_MAX_GEMM_GROUP];
#pragma unroll
    for (int i = 0; i < JAMIE_MAX_GEMM_GROUP; ++i) ntl[i] = g.ntiles[i];
    bool found = false;
#pragma unroll
    for (int i = 0; i < JAMIE_MAX_GEMM_GROUP; ++i) {
        const int T = ntl[i], qp = T >> 3, rp = T & 7;
        const int j = (xcd - rot) & 7;
        const int cp = qp + (j < rp ? 1 : 0);
        const bool hit = !found && slot < cp;
        pi = hit ? i : pi;
        t = hit ? j * qp + min(j, rp) + slot : t;
        slot = (found || hit) ? slot : slot - cp;
        found = found || hit;
        rot = (rot + rp) & 7;
    }
    const GemmBDev P = g.p[pi];          // (by value: every field in one burst of scalar loads, not a round trip per first use)
    const int tm_i = t % P.tiles_m;
    const int tn_i = (t / P.tiles_m) % P.tiles_n;
    const int ks = t / (P.tiles_m * P.tiles_n);
    const int m0 = tm_i * BM, n0 = tn_i * BN;
    const int kbeg = ks * P.kchunk;
    const int kend = min(P.K, kbeg + P.kchunk);
    const int nk = (kend - kbeg + BK - 1) / BK;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm0 = (wid / WN) * (TM * 32), wn0 = (wid % WN) * (TN * 32);
    const int r = lane & 31, h = lane >> 5;

    const __amdgpu_buffer_rsrc_t a_rs = __builtin_amdgcn_make_buffer_rsrc((void*)P.A, 0, (int)P.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rs = __builtin_amdgcn_make_buffer_rsrc((void*)P.B, 0, (int)P.b_bytes, 0x00020000);
    unsigned a_off[LA], b_off[LB];
    int a_k[LA], b_k[LB], a_lds[LA], b_lds[LB];
#pragma unroll
    for (int j = 0; j < LA; ++j) {
        const int f = tid + j * NT, row = f / CH, c = f % CH;
        const int gm = m0 + row;
        a_off[j] = gm < P.M ? ((unsigned)gm * (unsigned)P.lda + (unsigned)c * 8u) * 2u : JB_OOB;
        a_k[j] = c * 8;
        a_lds[j] = row * LDB + c * 16;
    }
#pragma unroll
    for (int j = 0; j < LB; ++j) {
        const int f = tid + j * NT, row = f / CH, c = f % CH;
        const int gn = n0 + row;
        b_off[j] = gn < P.N ? ((unsigned)gn * (unsigned)P.ldb + (unsigned)c * 8u) * 2u : JB_OOB;
        b_k[j] = c * 8;
        b_lds[j] = row * LDB + c * 16;
    }

    u32x4 ra[D][LA], rb[D][LB];
#define JB_LOAD(ST, K0)                                                                                              \
    {                                                                                                                 \
        _Pragma("unroll") for (int j = 0; j < LA; ++j) {                                                              \
            const bool ok = a_off[j] != JB_OOB && (K0) + a_k[j] < kend;                                               \
            ra[ST][j] = __builtin_amdgcn_raw_buffer_load_b128(a_rs, ok ? (int)(a_off[j] + (unsigned)(K0) * 2u) : (int)JB_OOB, 0, 0); \
        }                                                                                                             \
        _Pragma("unroll") for (int j = 0; j < LB; ++j) {                                                              \
            const bool ok = b_off[j] != JB_OOB && (K0) + b_k[j] < kend;                                               \
            rb[ST][j] = __builtin_amdgcn_raw_buffer_load_b128(b_rs, ok ? (int)(b_off[j] + (unsigned)(K0) * 2u) : (int)JB_OOB, 0, 0); \
        }                                                                                                             \
    }
#define JB_STORE(ST, BUF)                                                                                            \
    {                                                                                                                 \
        unsigned char* As_ = smem + (BUF) * (A_SZ + B_SZ);                                                            \
        unsigned char* Bs_ = As_ + A_SZ;                                                                              \
        _Pragma("unroll") for (int j = 0; j < LA; ++j) *reinterpret_cast<u32x4*>(As_ + a_lds[j]) = ra[ST][j];        \
        _Pragma("unroll") for (int j = 0; j < LB; ++j) *reinterpret_cast<u32x4*>(Bs_ + b_lds[j]) = rb[ST][j];        \
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // prologue: tiles 0 .. D-1 in flight; tile 0 to LDS
#pragma unroll
    for (int u = 0; u < D; ++u)
        if (u < nk) JB_LOAD(u, kbeg + u * BK)
    if (nk > 0) JB_STORE(0, 0)
    __syncthreads();
    for (int kt0 = 0; kt0 < nk; kt0 += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const int kt = kt0 + u;
            if (kt < nk) {
                const int cur = kt & 1;
                // register stage u held tile kt (already in LDS): refill it with tile kt + D
                if (kt + D < nk && JAMIE_GEMMB_ABL != 1) JB_LOAD(u, kbeg + (kt + D) * BK)
                const unsigned char* As = smem + cur * (A_SZ + B_SZ);
                const unsigned char* Bs = As + A_SZ;
                bf16x8 af[2][TM], bf[2][TN];
                auto read_frags = [&](int buf, int s) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
                        af[buf][i] = *reinterpret_cast<const bf16x8*>(As + (wm0 + i * 32 + r) * LDB + s * 32 + h * 16);
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        bf[buf][j] = *reinterpret_cast<const bf16x8*>(Bs + (wn0 + j * 32 + r) * LDB + s * 32 + h * 16);
                };
                read_frags(0, 0);
#pragma unroll
                for (int s = 0; s < BK / 16; ++s) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (s + 1 < BK / 16 && JAMIE_GEMMB_ABL != 4) read_frags((s + 1) & 1, s + 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                        {
#if JAMIE_GEMMB_ABL == 2
                            asm volatile("" ::"v"(af[s & 1][i]), "v"(bf[s & 1][j]));
#else
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[(JAMIE_GEMMB_ABL == 4 ? 0 : s) & 1][i], bf[(JAMIE_GEMMB_ABL == 4 ? 0 : s) & 1][j], acc[i][j], 0, 0, 0);
#endif
                        }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (kt + 1 < nk && JAMIE_GEMMB_ABL != 3) JB_STORE((u + 1) % D, cur ^ 1)     // tile kt+1 lives in register stage (u+1) % D
                __syncthreads();
            }
        }
    }
#undef JB_LOAD
#undef JB_STORE

    // ---- epilogue (C/D map: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)) ----
    float* Cout = P.C + (long long)ks * P.slab_stride;
    const bool add_bias = (P.bias != nullptr) && ks == 0;
    float local = 0.f;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn0 + j * 32 + r;
        if (n >= P.N) continue;
        const float bv = add_bias ? P.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (m >= P.M) continue;
                float v = acc[i][j][e] + bv;
                float* cp = Cout + (long long)m * P.ldc + n;
                if (P.epi == JAMIE_EPI_STORE) {
                    if (P.accumulate) v += *cp;
                    *cp = v;
                } else {  // JAMIE_EPI_MSE
                    const float d = v - P.aux0[(long long)m * P.aux_ld + n];
                    local += d * d;
                    *cp = d * P.scale;
                }
            }
        }
    }
    if (P.epi == JAMIE_EPI_MSE && P.partial != nullptr) {
        const float tot = block_sum(local, red);
        if (tid == 0) P.partial[t] = tot * P.pscale;
    }
}


// ------------------------------------------------------------------------------------------------
// LDS-DMA variant (BK = 64).  Ablations of the register-staged kernel above (tools/ablate_gemm_bf16.sh) show that
// the global -> VGPR -> ds_write_b128 staging path, not the MFMAs or the fragment reads, sets its time (32 us ->
// 16 us without it).  Here full k-tiles go global -> LDS directly (`global_load_lds_dwordx4`, 1 KiB = 8 rows x 128 B
// per wave-instruction, no VGPRs, no ds_write).  The LDS image must then be lane-linear, so rows cannot be padded:
// bank conflicts are avoided by an XOR swizzle applied on the SOURCE address and on the fragment read
// (cdna_hip_programming.md rule 21): physical 16-byte chunk = logical chunk ^ ((row >> 1) & 7); with 128-byte rows
// the 16 rows of every ds_read_b128 lane group then hit 16 distinct slots of the 256-byte bank line.
// Rows beyond M / N re-read the last valid row (their outputs are never stored); a partial last k-tile is staged
// through registers with masked buffer loads into the same swizzled image.
// ------------------------------------------------------------------------------------------------
// NB LDS buffers: the DMA of tiles kt+1 .. kt+NB-1 is in flight while tile kt is multiplied.  With more than one tile
// in flight `__syncthreads()` must go (its fence drains vmcnt(0)): raw s_barrier + hand-counted `s_waitcnt vmcnt(N)`,
// N = 4 glds per wave x DMA tiles younger than the one about to be read (cdna_hip_programming.md 'Pipelining across
// barriers').  One barrier per k-step: it publishes tile kt (every wave waited for its own pieces) and frees buffer
// (kt-1) % NB, into which tile kt+NB-1 is then issued.
template <int BM, int BN, int WM, int WN, int TAG, int NB, int ILV>
__global__ __launch_bounds__(WM * WN * 64) void gemm_bf16_dma_kernel(GemmBGroup g) {
    constexpr int BK = 64, NW = WM * WN, NT = NW * 64;
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int A_SZ = BM * 128, B_SZ = BN * 128;      // bytes, unpadded 128-byte rows
    constexpr int PA = BM / 8 / NW, PB = BN / 8 / NW;    // 1-KiB pieces per wave
    constexpr int LA = BM * 8 / NT, LB = BN * 8 / NT;    // 16-byte chunks per thread (register tail path)
    static_assert(PA >= 1 && PB >= 1 && (BM / 8) % NW == 0 && (BN / 8) % NW == 0, "tile/wave mismatch");
    // ONE __shared__ object: a second one beside an LDS-DMA staging array makes hipcc drain vmcnt(0) before the
    // first ds_read of every k-step (cdna_hip_programming.md, 'Three .s-level traps' (a)); `red` is carved from it
    // (over buffer 0: it is only used after the last k-step's barrier), so NB x 32 KiB tiles can fill the 160 KiB
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NB * (A_SZ + B_SZ)];
    float* red = reinterpret_cast<float*>(smem);

    // ---- block -> (problem, tile): per-problem XCD chunks (see gemm_f32.hip) ----
    const int bid = blockIdx.x;
    const int xcd = bid & 7;
    int slot = bid >> 3;
    int pi = 0, t = 0, rot = 0;
    // (branch-free, every problem's tile count loaded up front -- unused problems hold 0; as a loop of guarded iterations this was one
    //  dependent scalar-memory round trip and three branches per problem in front of the first load: gemm_f32.hip)
    int ntl[JAMIE_MAX_GEMM_GROUP];
#pragma unroll
    for (int i = 0; i < JAMIE_MAX_GEMM_GROUP; ++i) ntl[i] = g.ntiles[i];
    bool found = false;
#pragma unroll
    for (int i = 0; i < JAMIE_MAX_GEMM_GROUP; ++i) {
        const int T = ntl[i], qp = T >> 3, rp = T & 7;
        const int j = (xcd - rot) & 7;
        const int cp = qp + (j < rp ? 1 : 0);
        const bool hit = !found && slot < cp;
        pi = hit ? i : pi;
        t = hit ? j * qp + min(j, rp) + slot : t;
        slot = (found || hit) ? slot : slot - cp;
        found = found || hit;
        rot = (rot + rp) & 7;
    }
    const GemmBDev P = g.p[pi];          // (by value: every field in one burst of scalar loads, not a round trip per first use)
    const int tm_i = t % P.tiles_m;
    const int tn_i = (t / P.tiles_m) % P.tiles_n;
    const int ks = t / (P.tiles_m * P.tiles_n);
    const int m0 = tm_i * BM, n0 = tn_i * BN;
    const int kbeg = ks * P.kchunk;
    const int kend = min(P.K, kbeg + P.kchunk);
    const int nk = (kend - kbeg + BK - 1) / BK;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm0 = (wid / WN) * (TM * 32), wn0 = (wid % WN) * (TN * 32);
    const int r = lane & 31, h = lane >> 5;


    const int lrow = lane >> 3, pch = lane & 7;
    const unsigned short* a_src[PA]; const unsigned short* b_src[PB];
    int a_dst[PA], b_dst[PB];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int pc = wid + NW * i, row = 8 * pc + lrow;
        const int gm = min(m0 + row, P.M - 1);
        a_src[i] = P.A + (long long)gm * P.lda + ((pch ^ ((row >> 1) & 7)) * 8);
        a_dst[i] = pc * 1024;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        const int pc = wid + NW * i, row = 8 * pc + lrow;
        const int gn = min(n0 + row, P.N - 1);
        b_src[i] = P.B + (long long)gn * P.ldb + ((pch ^ ((row >> 1) & 7)) * 8);
        b_dst[i] = pc * 1024;
    }
    const int nfull = (kend - kbeg) / BK;
    typedef const void __attribute__((address_space(1)))* gptr_t;
    typedef void __attribute__((address_space(3)))* lptr_t;
    // part < 0: the whole tile; part = 0..3 (ILV): the quarter of this wave's pieces issued inside MFMA sub-step `part`
    auto stage = [&](int buf, int kt, int part) {
        unsigned char* As = smem + buf * (A_SZ + B_SZ);
        unsigned char* Bs = As + A_SZ;
        const int k0 = kbeg + kt * BK;
        if (kt < nfull) {
#pragma unroll
            for (int i = 0; i < PA; ++i)
                if (part < 0 || (i & 3) == part)
                    __builtin_amdgcn_global_load_lds((gptr_t)(a_src[i] + k0), (lptr_t)(As + a_dst[i]), 16, 0, 0);
#pragma unroll
            for (int i = 0; i < PB; ++i)
                if (part < 0 || ((i + PA) & 3) == part)
                    __builtin_amdgcn_global_load_lds((gptr_t)(b_src[i] + k0), (lptr_t)(Bs + b_dst[i]), 16, 0, 0);
        } else if (part <= 0) {   // partial k-tile: masked loads through registers into the same swizzled image
#pragma unroll
            for (int j = 0; j < LA; ++j) {
                const int f = tid + j * NT, row = f >> 3, c = f & 7;
                const int gm = min(m0 + row, P.M - 1);
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (k0 + c * 8 < kend) v = *reinterpret_cast<const uint4*>(P.A + (long long)gm * P.lda + k0 + c * 8);
                *reinterpret_cast<uint4*>(As + row * 128 + ((c ^ ((row >> 1) & 7)) << 4)) = v;
            }
#pragma unroll
            for (int j = 0; j < LB; ++j) {
                const int f = tid + j * NT, row = f >> 3, c = f & 7;
                const int gn = min(n0 + row, P.N - 1);
                uint4 v = make_uint4(0u, 0u, 0u, 0u);
                if (k0 + c * 8 < kend) v = *reinterpret_cast<const uint4*>(P.B + (long long)gn * P.ldb + k0 + c * 8);
                *reinterpret_cast<uint4*>(Bs + row * 128 + ((c ^ ((row >> 1) & 7)) << 4)) = v;
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int swz = (r >> 1) & 7;
    constexpr int GL = PA + PB;                       // glds instructions per wave and tile
    static_assert(GL * (NB - 1) <= 60 && NB <= 6, "vmcnt immediate");
    // prologue: tiles 0 .. NB-2 in flight
#pragma unroll
    for (int u = 0; u < NB - 1; ++u)
        if (u < nk) stage(u, u, -1);
    for (int kt = 0; kt < nk; ++kt) {
        // wait for MY pieces of tile kt: DMA tiles younger than kt that are already issued = tiles kt+1 .. kt+NB-2
        // (only full tiles are DMA; a partial last tile went through registers and was waited for by the compiler)
        const int younger = max(0, min(kt + NB - 2, nfull - 1) - kt);
        if (kt >= nfull || younger == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GL) : "memory");
        else if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * GL) : "memory");
        else if (younger == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NB > 4 ? 3 * GL : 0) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NB > 5 ? 4 * GL : 0) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const bool more = kt + NB - 1 < nk;
        if (!ILV && more && JAMIE_GEMMB_ABL != 7) stage((kt + NB - 1) % NB, kt + NB - 1, -1);
        const int cur = kt % NB;
        const unsigned char* As = smem + cur * (A_SZ + B_SZ);
        const unsigned char* Bs = As + A_SZ;
        bf16x8 af[2][TM], bf[2][TN];
        auto read_frags = [&](int buf, int s) {
            const int off = ((2 * s + h) ^ swz) << 4;
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[buf][i] = *reinterpret_cast<const bf16x8*>(As + (wm0 + i * 32 + r) * 128 + off);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bf[buf][j] = *reinterpret_cast<const bf16x8*>(Bs + (wn0 + j * 32 + r) * 128 + off);
        };
        read_frags(0, 0);
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            // the prefetch goes AFTER the first MFMA of the sub-step: hipcc puts an uncounted lgkmcnt(0) before that MFMA
            __builtin_amdgcn_sched_barrier(0);
            if (JAMIE_GEMMB_ABL != 6 || (kt == 0 && s == 0))
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s & 1][0], bf[s & 1][0], acc[0][0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (s + 1 < BK / 16 && JAMIE_GEMMB_ABL != 6) read_frags((s + 1) & 1, s + 1);
            if (ILV && more) stage((kt + NB - 1) % NB, kt + NB - 1, s);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    if (i + j > 0 && (JAMIE_GEMMB_ABL != 6 || (kt == 0 && s == 0)))
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s & 1][i], bf[s & 1][j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();

    // ---- epilogue (C/D map: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5)) ----
    float* Cout = P.C + (long long)ks * P.slab_stride;
    const bool add_bias = (P.bias != nullptr) && ks == 0;
    float local = 0.f;
#if JAMIE_GEMMB_ABL == 5
    {   // diagnostic: no output stores (one conditional store keeps the accumulators alive)
        float tot = 0.f;
        for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int e = 0; e < 16; ++e) tot += acc[i][j][e];
        if (tot == 123.456f) Cout[0] = tot;
        return;
    }
#endif
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn0 + j * 32 + r;
        if (n >= P.N) continue;
        const float bv = add_bias ? P.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
                if (m >= P.M) continue;
                float v = acc[i][j][e] + bv;
                float* cp = Cout + (long long)m * P.ldc + n;
                if (P.epi == JAMIE_EPI_STORE) {
                    if (P.accumulate) v += *cp;
                    *cp = v;
                } else {  // JAMIE_EPI_MSE
                    const float d = v - P.aux0[(long long)m * P.aux_ld + n];
                    local += d * d;
                    *cp = d * P.scale;
                }
            }
        }
    }
    if (P.epi == JAMIE_EPI_MSE && P.partial != nullptr) {
        const float tot = block_sum(local, red);
        if (tid == 0) P.partial[t] = tot * P.pscale;
    }
}


// ------------------------------------------------------------------------------------------------
// Second LDS-DMA kernel, for the LARGE tiles (256x128 / 128x128 with 64x64 per wave).  Ablations of the kernel above
// (tools/ablate_gemm_bf16_epi.sh: DMA only / MFMA only / no stores) show (a) global -> LDS moves ~27 B/clk/CU
// (62-66 GB/s per CU, ~17 TB/s chip-wide) whatever the tile, so the time of a product is (A + B bytes pulled into the
// CUs) / 17 TB/s: only a larger tile lowers it (256x128 pulls 3/8 of the bytes of 64x64); (b) with one barrier at the
// top of every k-step the first fragment reads of a tile have no MFMA to hide behind (8 waves: 1570 instead of 1024
// cycles per k-step); (c) the dword stores of the C/D map (2 x 128 B per instruction) cost 7 us for a 256x128 tile.
// Hence: the barrier that publishes tile kt+1 sits inside the LAST sub-step of tile kt, followed by the DMA issue of
// tile kt+NB and the first fragment reads of tile kt+1, all of it covered by that sub-step's MFMAs; and the MFMA
// operands are swapped (D = W-fragment x a-fragment), which transposes the accumulator map: a lane then holds 4
// CONSECUTIVE n for one m, stored as one 16-byte access (4x fewer store instructions).
// TRM = 0: no problem of the launch has a k-row-major operand (forward launches): the a_tr / b_tr paths are compiled out,
// TRM = 1: per-problem flags (backward launches: dX reads W as stored, dW reads dy and a as stored)
// (Round 5, measured neutral and taken out again: the eight tile counts as leading scalar kernel arguments delivered in SGPRs by
//  hipcc's kernarg preload, -mllvm -amdgpu-kernarg-preload-count=8, so that the block decode starts without a scalar-memory
//  round trip: 547.3 against 547.2 us per step, profiles/r05_ab_kernarg_preload_neutral.log -- the descriptor's own load behind
//  the decode is the round trip that counts.)
template <int BM, int BN, int WM, int WN, int TAG, int NB, int TRM, bool FUSE = false>
__device__ __forceinline__ void gemm_bf16_dma2_body(const GemmBGroup& g, const BnFuse* fz = nullptr) {
    constexpr int BK = 64, NW = WM * WN, NT = NW * 64;
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int A_SZ = BM * 128, B_SZ = BN * 128, T_SZ = A_SZ + B_SZ;
    constexpr int PA = BM / 8 / NW, PB = BN / 8 / NW;    // 1-KiB pieces per wave
    constexpr int LA = BM * 8 / NT, LB = BN * 8 / NT;    // 16-byte chunks per thread (register tail path)
    constexpr int GL = PA + PB;
    static_assert(PA >= 1 && PB >= 1 && (BM / 8) % NW == 0 && (BN / 8) % NW == 0, "tile/wave mismatch");
    static_assert(GL * (NB - 1) <= 63 && NB >= 2 && NB <= 5, "vmcnt immediate");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NB * T_SZ];
    float* red = reinterpret_cast<float*>(smem);

    JB_STAMP(0);
    const int bid = blockIdx.x;
    const int xcd = bid & 7;
    int slot = bid >> 3;
    int pi = 0, t = 0, rot = 0;
    // (branch-free, every problem's tile count loaded up front -- unused problems hold 0; as a loop of guarded iterations this was one
    //  dependent scalar-memory round trip and three branches per problem in front of the first load: gemm_f32.hip)
    int ntl[JAMIE_MAX_GEMM_GROUP];
#pragma unroll
    for (int i = 0; i < JAMIE_MAX_GEMM_GROUP; ++i) ntl[i] = g.ntiles[i];
    bool found = false;
#pragma unroll
    for (int i = 0; i < JAMIE_MAX_GEMM_GROUP; ++i) {
        const int T = ntl[i], qp = T >> 3, rp = T & 7;
        const int j = (xcd - rot) & 7;
        const int cp = qp + (j < rp ? 1 : 0);
        const bool hit = !found && slot < cp;
        pi = hit ? i : pi;
        t = hit ? j * qp + min(j, rp) + slot : t;
        slot = (found || hit) ? slot : slot - cp;
        found = found || hit;
        rot = (rot + rp) & 7;
    }
    const GemmBDev P = g.p[pi];          // (by value: every field in one burst of scalar loads, not a round trip per first use)
    // FUSE: the tiles_m x splitk workgroups of one column strip are adjacent in the tile list (same XCD chunk, dispatched
    // together): they hand their slabs to each other inside the launch
    const int pn = P.tiles_m * P.splitk;
    // Tile order inside a K slice: M-tiles in groups of JB_MG, the N-tiles of a group before the next group, so that the contiguous
    // run of tiles an XCD gets (its chunk of the list) is a BLOCK of the output -- with 32 x 16 tiles (dW of a 2d x d layer: 64 per
    // XCD) 8 x 8 tiles on 16 operand panels, each read 8 times through that XCD's L2, instead of 32 x 2 tiles on 34 panels (4.3 MB:
    // more than the 4 MB L2).  Launches with up to JB_MG M-tiles (forward, dX: M = batch) keep their order.  Same arithmetic.
    constexpr int JB_MG = 8;
    int tm_b, tn_b;
    {
        const int tt = t % (P.tiles_m * P.tiles_n);
        const int ng = P.tiles_m / JB_MG, rem = P.tiles_m - ng * JB_MG, gsz = JB_MG * P.tiles_n;
        const int gi = tt / gsz;
        if (gi < ng) {
            const int w_ = tt - gi * gsz;
            tn_b = w_ / JB_MG; tm_b = gi * JB_MG + w_ % JB_MG;
        } else {
            const int w_ = tt - ng * gsz;
            tn_b = w_ / rem; tm_b = ng * JB_MG + w_ % rem;
        }
    }
    const int tm_i = FUSE ? (t % pn) % P.tiles_m : tm_b;
    const int tn_i = FUSE ? t / pn : tn_b;
    const int ks = FUSE ? (t % pn) / P.tiles_m : t / (P.tiles_m * P.tiles_n);
    const int t_id = tm_i + P.tiles_m * tn_i + ks * P.tiles_m * P.tiles_n;       // tile id of the per-tile partial sums: m_tile + tiles_m * n_tile
    const int m0 = tm_i * BM, n0 = tn_i * BN;
    const int kbeg = ks * P.kchunk;
    const int kend = min(P.K, kbeg + P.kchunk);
    const int nk = (kend - kbeg + BK - 1) / BK;
    const int nfull = (kend - kbeg) / BK;
    JB_STAMPV(4, pi * 1000 + nk);
    JB_STAMPV(5, __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) * 1000 + __builtin_amdgcn_s_getreg(((8 - 1) << 11) | (8 << 6) | 4));   // XCC_ID, HW_ID cu/se bits

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm0 = (wid / WN) * (TM * 32), wn0 = (wid % WN) * (TN * 32);
    const int r = lane & 31, h = lane >> 5;
    const int lrow = lane >> 3, pch = lane & 7;
    const unsigned short* a_src[PA]; const unsigned short* b_src[PB];
    const bool a_tr = TRM != 0 && P.a_tr != 0;
    const long long a_kstep = a_tr ? (long long)P.lda : 1;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        if (a_tr) {      // A stored [K, M]: the same [k][m] image as the b_tr operand (below)
            const int piece = wid + NW * i, krow = 4 * piece + (lane >> 4);
            const int lc = (lane & 15) ^ (((krow & 3) << 2) | ((krow >> 2) & 3));
            a_src[i] = P.A + (long long)krow * P.lda + min(m0 + lc * 8, P.M - 8);
        } else {
            const int row = 8 * (wid + NW * i) + lrow;
            a_src[i] = P.A + (long long)min(m0 + row, P.M - 1) * P.lda + ((pch ^ ((row >> 1) & 7)) * 8);
        }
    }
    // B operand as stored: [N, K] with K contiguous (forward, dW) or, b_tr, [K, N] with N contiguous (dX = dy W reads
    // the weights W [out, in] as they are: no transposed copy).  The b_tr tile is [64 k][BN n] with 256-byte rows, one
    // 1-KiB DMA piece = 4 k-rows, 16-byte chunks XOR-swizzled by ((k & 3) << 2) | ((k >> 2) & 3): the image on which
    // both DMA fills and the 32x32x16 transposed reads are conflict-free (cdna_hip_programming.md T10 (b)).
    const bool b_tr = TRM != 0 && P.b_tr != 0;
    const long long b_kstep = b_tr ? (long long)P.ldb : 1;
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        if (b_tr) {
            const int piece = wid + NW * i, krow = 4 * piece + (lane >> 4);
            const int lc = (lane & 15) ^ (((krow & 3) << 2) | ((krow >> 2) & 3));
            b_src[i] = P.B + (long long)krow * P.ldb + min(n0 + lc * 8, P.N - 8);
        } else {
            const int row = 8 * (wid + NW * i) + lrow;
            b_src[i] = P.B + (long long)min(n0 + row, P.N - 1) * P.ldb + ((pch ^ ((row >> 1) & 7)) * 8);
        }
    }
    typedef const void __attribute__((address_space(1)))* gptr_t;
    typedef void __attribute__((address_space(3)))* lptr_t;
    auto stage = [&](int buf, int kt) {
        unsigned char* As = smem + buf * T_SZ;
        unsigned char* Bs = As + A_SZ;
        const int k0 = kbeg + kt * BK;
        if (kt < nfull) {
#pragma unroll
            for (int i = 0; i < PA; ++i)
                __builtin_amdgcn_global_load_lds((gptr_t)(a_src[i] + k0 * a_kstep), (lptr_t)(As + (wid + NW * i) * 1024), 16, 0, 0);
#pragma unroll
            for (int i = 0; i < PB; ++i)
                __builtin_amdgcn_global_load_lds((gptr_t)(b_src[i] + k0 * b_kstep), (lptr_t)(Bs + (wid + NW * i) * 1024), 16, 0, 0);
        } else {
            // partial last k-tile: the SAME LDS-DMA fill, with the lanes whose 16-byte chunk lies at k >= kend reading a zero
            // constant instead (the source address of a DMA is per lane; K is a multiple of 8, so a chunk is in or out whole).
            // Rounds 1-3 staged this tile through registers: plain loads beside pending LDS-DMA make hipcc wait vmcnt(0), i.e.
            // the ring drained once per tile, on the workgroups whose K slice is the ragged one (the launch's critical path).
#pragma unroll
            for (int i = 0; i < PA; ++i) {
                const int piece = wid + NW * i;
                const int kof = a_tr ? 4 * piece + (lane >> 4) : ((pch ^ (((8 * piece + lrow) >> 1) & 7)) * 8);
                const unsigned short* src = (k0 + kof < kend) ? a_src[i] + k0 * a_kstep : reinterpret_cast<const unsigned short*>(jb_zero16);
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(As + piece * 1024), 16, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < PB; ++i) {
                const int piece = wid + NW * i;
                const int kof = b_tr ? 4 * piece + (lane >> 4) : ((pch ^ (((8 * piece + lrow) >> 1) & 7)) * 8);
                const unsigned short* src = (k0 + kof < kend) ? b_src[i] + k0 * b_kstep : reinterpret_cast<const unsigned short*>(jb_zero16);
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Bs + piece * 1024), 16, 0, 0);
            }
        }
    };
    // my pieces of tile `tile` have landed; `last` = youngest tile issued so far (every tile is GL LDS-DMA instructions per wave)
    auto wait_tile = [&](int tile, int last) {
        const int younger = max(0, min(last, nk - 1) - tile);
        if (younger == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GL) : "memory");
        else if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NB > 2 ? 2 * GL : 0) : "memory");
        else if (younger == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NB > 3 ? 3 * GL : 0) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NB > 4 ? 4 * GL : 0) : "memory");
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int swz = (r >> 1) & 7;
    bf16x8 af[2][TM], bf[2][TN];
    s16x4 afl[2][TM], afh[2][TM], bfl[2][TN], bfh[2][TN];      // k-row-major operands: the halves of a fragment as the asm reads wrote them
    // transposed read of the [k][n] image: in a 16-lane group lane 4q+p supplies row q, columns 4p..4p+3 of a 4 x 16
    // block and lane i receives column i of the 4 rows; lane (n = lane & 31, kh = lane >> 5) of the MFMA operand needs
    // k = 16 s + 8 kh + 0..7 of column n: two reads (k-rows +0..3, +4..7).  Row 16 s + 8 kh + 4 t + q has row & 3 = q
    // and (row >> 2) & 3 = (2 kh + t) & 3 in the swizzle.
    const int tq = (lane >> 2) & 3, tp = lane & 3, tc0 = wn0 + 16 * ((lane >> 4) & 1), ta0 = wm0 + 16 * ((lane >> 4) & 1);
    typedef s16x4 __attribute__((address_space(3)))* trp_t;
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    auto read_frags = [&](const unsigned char* As, int fb, int s, auto tr, auto tra) {
        const unsigned char* Bs = As + A_SZ;
        const int off = ((2 * s + h) ^ swz) << 4;
        if constexpr (decltype(tra)::value) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int ch = ((ta0 + 32 * i) >> 3) + (tp >> 1);
                const unsigned char* base = As + s * (16 * BM * 2) + 8 * (tp & 1);
                const int o0 = (8 * h + tq) * (BM * 2) + ((ch ^ ((tq << 2) | ((2 * h) & 3))) << 4);
                const int o1 = (8 * h + 4 + tq) * (BM * 2) + ((ch ^ ((tq << 2) | ((2 * h + 1) & 3))) << 4);
                afl[fb][i] = jb_ds_read_tr16(base + o0);          // (concatenated behind the wait: take_frags)
                afh[fb][i] = jb_ds_read_tr16(base + o1);
            }
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i)
                af[fb][i] = *reinterpret_cast<const bf16x8*>(As + (wm0 + i * 32 + r) * 128 + off);
        }
        if constexpr (decltype(tr)::value) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                // the next 32 columns are 4 chunks on: XOR with 4 flips bit 2 of the chunk index, which the swizzle
                // term ((q << 2) | ...) may also flip, so the offset is recomputed per j from the chunk index
                const int ch = ((tc0 + 32 * j) >> 3) + (tp >> 1);
                const unsigned char* base = Bs + s * (16 * BN * 2) + 8 * (tp & 1);
                const int o0 = (8 * h + tq) * (BN * 2) + ((ch ^ ((tq << 2) | ((2 * h) & 3))) << 4);
                const int o1 = (8 * h + 4 + tq) * (BN * 2) + ((ch ^ ((tq << 2) | ((2 * h + 1) & 3))) << 4);
                bfl[fb][j] = jb_ds_read_tr16(base + o0);
                bfh[fb][j] = jb_ds_read_tr16(base + o1);
            }
        } else {
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bf[fb][j] = *reinterpret_cast<const bf16x8*>(Bs + (wn0 + j * 32 + r) * 128 + off);
        }
    };

    // the hand-written wait for the asm reads of fragment set `fb`, naming every half it retires, then their concatenation into
    // the MFMA operands (see jb_ds_read_tr16)
    auto take_frags = [&](int fb, auto tr, auto tra) {
        if constexpr (decltype(tr)::value) {
            jb_lds_wait_for(bfl[fb][0]);
            jb_after_wait(bfh[fb][0]);
#pragma unroll
            for (int j = 1; j < TN; ++j) { jb_after_wait(bfl[fb][j]); jb_after_wait(bfh[fb][j]); }
            if constexpr (decltype(tra)::value) {
#pragma unroll
                for (int i = 0; i < TM; ++i) { jb_after_wait(afl[fb][i]); jb_after_wait(afh[fb][i]); }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j)
                bf[fb][j] = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(bfl[fb][j], bfh[fb][j], 0, 1, 2, 3, 4, 5, 6, 7));
            if constexpr (decltype(tra)::value) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    af[fb][i] = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(afl[fb][i], afh[fb][i], 0, 1, 2, 3, 4, 5, 6, 7));
            }
        }
    };
    // prologue: every buffer in flight, then tile 0 published and its first fragments read
    if constexpr (!FUSE) JB_STAMP(6);
#ifdef JB_PRO_BARRIER
    // A/B (round 5): tile 0's pieces of EVERY wave go out before any wave's tile 1 / 2 pieces -- the waves reach this point up to
    // ~0.5 us apart, and an early wave's later tiles otherwise queue in front of a late wave's tile-0 pieces (tile 0 is
    // published when its LAST piece lands)
    if (nk > 0) stage(0, 0);
    if (NB > 1 && nk > 1) {
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int u = 1; u < NB; ++u)
            if (u < nk) stage(u, u);
    }
#else
#pragma unroll
    for (int u = 0; u < NB; ++u)
        if (u < nk) stage(u, u);
#endif
    if constexpr (!FUSE) JB_STAMP(7);
    wait_tile(0, min(NB, nk) - 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    JB_STAMP(1);
    auto k_loop = [&](auto tr, auto tra) {
    if (nk <= 0) return;        // (an empty K slice stores zeros; no asm read may be left pending into the epilogue: tools/isa_check.py)
    read_frags(smem, 0, 0, tr, tra);
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const unsigned char* As = smem + cur * T_SZ;
        const int nxt = (cur + 1 == NB) ? 0 : cur + 1;
#pragma unroll
        for (int s = 0; s < BK / 16; ++s) {
            // hipcc waits lgkmcnt(0) (not a counted wait) before the first MFMA of a sub-step while LDS-DMA is
            // pending, so the prefetch of the next fragments is issued AFTER that MFMA: the wait then only covers
            // reads that have had a whole sub-step to land
            take_frags(s & 1, tr, tra);        // (the asm transposed reads: no compiler-side tracking -- wait + concatenate)
            __builtin_amdgcn_sched_barrier(0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[s & 1][0], af[s & 1][0], acc[0][0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (s + 1 < BK / 16) {
                read_frags(As, (s + 1) & 1, s + 1, tr, tra);
            } else if (kt + 1 < nk) {
                // tile kt+1 becomes visible and buffer `cur` free (every wave holds its last fragments of tile kt in
                // registers: lgkmcnt(0)); tile kt+NB goes into it and the first fragments of tile kt+1 are fetched
                // under the MFMAs of this sub-step
                wait_tile(kt + 1, min(kt + NB - 1, nk - 1));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                if (kt + NB < nk) stage(cur, kt + NB);
                read_frags(smem + nxt * T_SZ, 0, 0, tr, tra);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    if (i + j > 0)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[s & 1][j], af[s & 1][i], acc[i][j], 0, 0, 0);
        }
        cur = nxt;
    }
    };
    if (a_tr) k_loop(std::true_type{}, std::true_type{});            // dW on the row-major activations / gradients
    else if (b_tr) k_loop(std::true_type{}, std::false_type{});      // dX on the weights as stored
    else k_loop(std::false_type{}, std::false_type{});
    // every asm read retired before the epilogue reuses a register (the loop's exit branch sits BEHIND the next tile's first reads
    // in program order: on that -- infeasible -- path hipcc knows of nothing pending; tools/isa_check.py walks it)
    jb_lds_wait();
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    JB_STAMP(2);

    // ---- epilogue: transposed C/D map -> m = lane & 31, n = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5) ----
    // Stored straight from that map a wave-instruction writes 32 rows x 32 bytes: 32 partial lines per instruction, and
    // the stores of a tile took 6.5-9 us (in-kernel stamps, tools/stamp_gemm_bf16.py) during which the workgroup keeps
    // its LDS and wave slots.  So every wave first turns its 32-row strips through a private LDS scratch (the k-loop's
    // buffers are free now) and then writes whole row segments: TN * 128 contiguous bytes per row, 64 / (TN * 8) rows
    // per instruction.
    float* Cout = P.C + (long long)ks * P.slab_stride;
    const bool add_bias = (P.bias != nullptr) && ks == 0;
    float local = 0.f;
    // scratch row stride in floats: 16 B pad (conflict-free b128 writes) wherever the LDS of the k-loop has room for it
    constexpr int SROW = TN * 32 + ((NB * T_SZ >= NW * 32 * (TN * 32 + 4) * 4) ? 4 : 0);
    constexpr int CPR = TN * 8, RPI = 64 / CPR;       // 16-byte chunks per row segment, rows per wave-instruction
    static_assert(NB * T_SZ >= NW * 32 * SROW * 4 && 64 % CPR == 0, "epilogue scratch");
    float* scr = reinterpret_cast<float*>(smem) + wid * (32 * SROW);
    const int cch = lane % CPR, rsub = lane / CPR;
    const int nc = n0 + wn0 + cch * 4;                // first of this lane's 4 output columns
    const bool v4 = P.vec && nc + 3 < P.N;
    float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (add_bias) {
        if (v4) b4 = *reinterpret_cast<const float4*>(P.bias + nc);
        else {
            if (nc < P.N) b4.x = P.bias[nc];
            if (nc + 1 < P.N) b4.y = P.bias[nc + 1];
            if (nc + 2 < P.N) b4.z = P.bias[nc + 2];
            if (nc + 3 < P.N) b4.w = P.bias[nc + 3];
        }
    }
    // Plain stores of whole interior tiles (split-K slabs, bf16 weight gradients: every large launch of the step) take a lean
    // path: the general loop below decides epilogue kind, accumulation, output type, cache policy and bounds per 16-byte piece
    // (~30 instructions and 6 branches each, 64-bit address arithmetic); here a piece is its bias add, (sum of squares,) pack and ONE
    // buffer store whose row offset is scalar -- the tile's CU slot is free again that much earlier (gemm_f32.hip, round 4).
    const unsigned c_elem = P.c_bf16 ? 2u : 4u;
    // panel layout (c_panel; fp32 only, host-checked): a row of a panel is 4 JAMIE_PANEL bytes and the panels of a slab follow one
    // another M rows apart -- the row pitch and the column term of the offset change, nothing else does
    const bool pan = !FUSE && P.c_panel != 0;
    constexpr unsigned PNW = JAMIE_PANEL, PNB = 4u * JAMIE_PANEL;          // panel width in columns / bytes per panel row
    static_assert(JAMIE_PANEL % 4 == 0 && (JAMIE_PANEL & (JAMIE_PANEL - 1)) == 0, "a panel is a power of two of at least 4 columns");
    const unsigned long long c_ext = pan ? (unsigned long long)((P.N + PNW - 1) / PNW) * (unsigned long long)P.M * PNB
                                         : ((unsigned long long)(P.M - 1) * (unsigned long long)P.ldc + (unsigned long long)P.N) * c_elem;
    if (!FUSE && P.epi == JAMIE_EPI_STORE && !P.accumulate && P.vec && (P.N & 3) == 0 && c_ext < 0xFFFFFFF0ull) {
        const __amdgpu_buffer_rsrc_t c_rs = __builtin_amdgcn_make_buffer_rsrc((void*)Cout, 0, (int)(unsigned)c_ext, 0x00020000);
        const unsigned ldcb = pan ? PNB : (unsigned)P.ldc * c_elem;
        const unsigned coff = pan ? ((unsigned)nc / PNW) * ((unsigned)P.M * PNB) + ((unsigned)nc % PNW) * 4u : (unsigned)nc * c_elem;
        // (columns beyond N: an out-of-range offset from the start -- N % 4 == 0, a 16-byte piece is in or out whole; rows beyond M:
        //  a select per piece, in the edge tiles' instance only.  The edge tiles matter: in a one-round launch the slowest tile ends it.)
        const unsigned voff = nc < P.N ? (unsigned)(m0 + wm0 + rsub) * ldcb + coff : 0xFFFFFFF0u;
        const bool edge_m = m0 + BM > P.M || n0 + BN > P.N;       // (an edge tile of either kind)
        auto emit = [&](auto bf_c, auto nt_c, auto edge_c) {
            constexpr bool C16 = decltype(bf_c)::value, NT_ST = decltype(nt_c)::value, EDGE = decltype(edge_c)::value;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<float4*>(scr + r * SROW + j * 32 + 8 * q + 4 * h) =
                            make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
#pragma unroll
                for (int rr = 0; rr < 32 / RPI; ++rr) {
                    const float4 a4 = *reinterpret_cast<const float4*>(scr + (rr * RPI + rsub) * SROW + cch * 4);
                    const float v0 = a4.x + b4.x, v1 = a4.y + b4.y, v2 = a4.z + b4.z, v3 = a4.w + b4.w;
                    unsigned vo = voff;
                    if (EDGE) vo = (m0 + wm0 + i * 32 + rr * RPI + rsub < P.M) ? voff : 0xFFFFFFF0u;
                    const float sq = v0 * v0 + v1 * v1 + v2 * v2 + v3 * v3;
                    if (EDGE) local += (vo != 0xFFFFFFF0u) ? sq : 0.f;          // (rows beyond M repeat row M - 1, columns beyond N hold other data)
                    else local += sq;
                    const unsigned so = (unsigned)(i * 32 + rr * RPI) * ldcb;
                    if (C16) {
                        auto bfr = [](float x) { return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)x); };
                        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                        u32x2 pk; pk.x = bfr(v0) | (bfr(v1) << 16); pk.y = bfr(v2) | (bfr(v3) << 16);
#ifdef JAMIE_STORE_SOFF       // (A/B: round 4's form, the row offset as the store's SCALAR offset)
                        __builtin_amdgcn_raw_buffer_store_b64(pk, c_rs, (int)vo, (int)so, NT_ST ? 2 : 0);
#else
                        // the row offset in the VECTOR offset for the 8-byte stores too (round 5): the ISA holds stores of <= 64 bits
                        // free of the store-data hazard, but the product now contains NO buffer store with an SGPR offset at all --
                        // what tests/test_isa_hazards.py asserts on the code objects, whatever a later toolchain schedules
                        __builtin_amdgcn_raw_buffer_store_b64(pk, c_rs, (int)((EDGE && vo == 0xFFFFFFF0u) ? vo : vo + so), 0, NT_ST ? 2 : 0);
#endif
                    } else {
                        // (the row offset in the VECTOR offset here: a 16-byte buffer store with a scalar-register offset had its data
                        //  registers overwritten under it -- wrong .y elements in fixed lanes, tools/debug_bf16_epilogue.py; hipcc's
                        //  hazard model holds that form free of the store-data hazard, gfx950 does not)
                        u32x4 pk; pk.x = __float_as_uint(v0); pk.y = __float_as_uint(v1); pk.z = __float_as_uint(v2); pk.w = __float_as_uint(v3);
                        __builtin_amdgcn_raw_buffer_store_b128(pk, c_rs, (int)(vo == 0xFFFFFFF0u ? vo : vo + so), 0, NT_ST ? 2 : 0);
                    }
                }
            }
        };
        auto emit2 = [&](auto bf_c, auto nt_c) {
            if (edge_m) { asm volatile("; rows beyond M"); emit(bf_c, nt_c, std::true_type{}); }
            else emit(bf_c, nt_c, std::false_type{});
        };
        if (P.c_bf16) {
            asm volatile("; bf16 output");
            if (P.store_nt) { asm volatile("; nt"); emit2(std::true_type{}, std::true_type{}); }
            else emit2(std::true_type{}, std::false_type{});
        } else {
            asm volatile("; fp32 output");
            if (P.store_nt) { asm volatile("; nt"); emit2(std::false_type{}, std::true_type{}); }
            else emit2(std::false_type{}, std::false_type{});
        }
    } else
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4*>(scr + r * SROW + j * 32 + 8 * q + 4 * h) =
                    make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
#pragma unroll
        for (int rr = 0; rr < 32 / RPI; ++rr) {
            const int row = rr * RPI + rsub;
            const float4 a4 = *reinterpret_cast<const float4*>(scr + row * SROW + cch * 4);
            const int m = m0 + wm0 + i * 32 + row;
            if (m >= P.M || nc >= P.N) continue;
            float* cp = Cout + (long long)m * P.ldc + nc;
            float v[4] = {a4.x + b4.x, a4.y + b4.y, a4.z + b4.z, a4.w + b4.w};
            if (v4) {
                if (P.epi == JAMIE_EPI_STORE) {
                    if (P.accumulate) {
                        const float4 o = *reinterpret_cast<const float4*>(cp);
                        v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w;
                    }
                    local += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];   // -> partial[t]: ||dW||^2 for the clip
                } else {  // JAMIE_EPI_MSE
                    const float4 x = *reinterpret_cast<const float4*>(P.aux0 + (long long)m * P.aux_ld + nc);
                    v[0] -= x.x; v[1] -= x.y; v[2] -= x.z; v[3] -= x.w;
                    local += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
                    v[0] *= P.scale; v[1] *= P.scale; v[2] *= P.scale; v[3] *= P.scale;
                }
                if (P.c_bf16) {       // bf16 output (weight gradients for jamie_clip_adam_g16): one 8-byte streaming store
                    auto bfr = [](float x) { return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)x); };
                    const unsigned long long pk = (unsigned long long)(bfr(v[0]) | (bfr(v[1]) << 16)) |
                                                  ((unsigned long long)(bfr(v[2]) | (bfr(v[3]) << 16)) << 32);
                    unsigned long long* c16 = reinterpret_cast<unsigned long long*>(
                        reinterpret_cast<unsigned short*>(Cout) + (long long)m * P.ldc + nc);
                    if (P.store_nt) __builtin_nontemporal_store(pk, c16);
                    else *c16 = pk;
                } else if (P.store_nt) {     // streamed past the caches: plain stores left 161 MB of dirty gradient lines per step in
                                      // L2 / Infinity Cache, whose write-back ran into the optimiser kernel (229 -> 208 us)
                    __builtin_nontemporal_store(v[0], cp); __builtin_nontemporal_store(v[1], cp + 1);
                    __builtin_nontemporal_store(v[2], cp + 2); __builtin_nontemporal_store(v[3], cp + 3);
                } else if constexpr (FUSE) {
                    // write-through (sc1): the slab is read by another workgroup of THIS launch (publish-large: no release
                    // fence, no dirty lines to write back before the ticket)
                    const __amdgpu_buffer_rsrc_t c_rs = __builtin_amdgcn_make_buffer_rsrc(
                        (void*)Cout, 0, (int)((unsigned)P.M * (unsigned)P.ldc * 4u), 0x00020000);
                    u32x4 pk;
                    pk.x = __float_as_uint(v[0]); pk.y = __float_as_uint(v[1]); pk.z = __float_as_uint(v[2]); pk.w = __float_as_uint(v[3]);
                    __builtin_amdgcn_raw_buffer_store_b128(pk, c_rs, (int)(((unsigned)m * (unsigned)P.ldc + (unsigned)nc) * 4u), 0, 16);
                } else {
                    *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (nc + e >= P.N) continue;
                    float w = v[e];
                    if (P.epi == JAMIE_EPI_STORE && P.c_bf16) {
                        (reinterpret_cast<unsigned short*>(Cout) + (long long)m * P.ldc + nc)[e] =
                            __builtin_bit_cast(unsigned short, (__bf16)w);
                        local += w * w;
                    } else if (P.epi == JAMIE_EPI_STORE) {
                        if (P.accumulate) w += cp[e];
                        cp[e] = w;
                        local += w * w;
                    } else {
                        const float d = w - P.aux0[(long long)m * P.aux_ld + nc + e];
                        local += d * d;
                        cp[e] = d * P.scale;
                    }
                }
            }
        }
    }
    // per-tile partial: the MSE term (EPI_MSE) or, for a plain store, the sum of squares of the stored values -- the
    // dW launches hand the gradient-norm kernel its partial sums (clip_grad_norm_, jamie.py:739) without a second pass
    if (P.partial != nullptr) {           // (block_sum's own barrier comes before it touches `red`, which the scratch overlaps)
        const float tot = block_sum(local, red);
        if (tid == 0) P.partial[t_id] = tot * P.pscale;
    }
#ifdef JAMIE_EXPERIMENTS
    if constexpr (FUSE) {
        // ---- hand-off (cdna_hip_programming.md, 'In-launch split-K reduction', sc1 form): every storing wave drains its
        // write-through stores, workgroup barrier, ONE lane takes the strip's ticket (relaxed, agent scope); the reducing
        // workgroup's lane 0 acquires at agent scope (drops this CU's L1 lines), drains, barrier, then the slab loads (sc1) ----
        const BnFuse& F = *fz;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        JB_STAMP(3);
        typedef __attribute__((address_space(1))) unsigned gu32;        // (global, never flat: the polled words)
        gu32* cnt = (gu32*)(F.tickets + 4 + 2 * (F.ticket_base[pi] + tn_i));
        int* sflag = reinterpret_cast<int*>(smem + 4096);
        if (tid == 0) *sflag = (int)__hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const int ticket = *sflag;
        constexpr int TEAMS = NT / 512;
        constexpr int UNITS = (BN / 16) / TEAMS;          // passes over the strip's 16-column sub-strips, TEAMS at a time
        static_assert(NT % 512 == 0 && (BN / 16) % TEAMS == 0, "teams of 512 threads");
        const bool every = F.mode == 2 && pn > 1;
        int u0 = 0, ustep = 1;
        if (every) {
            u0 = ticket; ustep = pn;
            if (u0 < UNITS && tid == 0) {                 // wait for the strip's other slices (bounded; co-resident by dispatch order)
                unsigned spins = 0;
                while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)pn) {
                    __builtin_amdgcn_s_sleep(8);
                    if (++spins > (1u << 22)) { __hip_atomic_store((gu32*)F.tickets, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                }
            }
        } else if (ticket != pn - 1) {
            return;                                       // (uniform) not the last arriver: done
        }
        JB_STAMP(6);
        if (u0 < UNITS) {
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
            const int team = tid >> 9, tl_ = tid & 511;
            float (*sh)[BN_CW] = reinterpret_cast<float (*)[BN_CW]>(smem + team * 512);
            for (int u = u0; u < UNITS; u += ustep) {
                const int col0 = n0 + 16 * (u * TEAMS + team);
                bn_fwd4_strip<4, 16>(F.p[pi], col0, tl_, col0 < P.N, sh, nullptr, F.p_drop, F.momentum, F.eps, F.slope, F.rng);
            }
        }
        // leave the counters zero for the next launch: the last arriver (mode 1) / the last to depart (mode 2: nobody is
        // still polling the arrivals word then)
        if (tid == 0) {
            if (!every) {
                __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if (__hip_atomic_fetch_add(cnt + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(pn - 1)) {
                __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(cnt + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        JB_STAMP(7);
        return;
    }
#endif
#ifdef JAMIE_GEMMB_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    JB_STAMP(3);
#endif
}

#ifdef JAMIE_EXPERIMENTS
// the large-tile kernel with the in-launch split-K reduction + BatchNorm forward (forward launches: no k-row-major operands)
template <int BM, int BN, int WM, int WN, int TAG, int NB>
__global__ __launch_bounds__(WM * WN * 64) void gemm_bf16_dma2_bn_kernel(GemmBGroup g, BnFuse f) {
    gemm_bf16_dma2_body<BM, BN, WM, WN, TAG, NB, 0, true>(g, &f);
}
#endif

template <int BM, int BN, int WM, int WN, int TAG, int NB, int TRM>
__global__ __launch_bounds__(WM * WN * 64) void gemm_bf16_dma2_kernel(GemmBGroup g) {
    gemm_bf16_dma2_body<BM, BN, WM, WN, TAG, NB, TRM>(g);
}

// The same launch with RIDERS: the workgroups after the `n_tiles` GEMM tiles take the gradient ranges no GEMM writes (their sums
// of squares for the clip norm) and the deferred finalisation of the latent backward pass (range_norm.h).  Used for the LAST dW
// launch of a backward pass, when every other gradient exists: the range-norm launch (10 us between the last GEMM and the
// optimiser) disappears into it.
template <int BM, int BN, int WM, int WN, int TAG, int NB, int TRM>
__global__ __launch_bounds__(WM * WN * 64) void gemm_bf16_dma2_ride_kernel(GemmBGroup g, RangeRide rr, int n_tiles) {
    if ((int)blockIdx.x >= n_tiles) {
        __shared__ float ride_red[(WM * WN + 1) * (SM_SLOTS + 2)];
        range_ride_block(rr, (int)blockIdx.x - n_tiles, ride_red);
        return;
    }
    gemm_bf16_dma2_body<BM, BN, WM, WN, TAG, NB, TRM>(g);
}

template <int BM, int BN, int BK, int WM, int WN, int D>
static int launch_b(const jamie_gemm_problem* pr, int count, hipStream_t st) {
    GemmBGroup g;
    memset(&g, 0, sizeof(g));
    g.count = count;
    int tiles = 0;
    bool big = true;
    for (int i = 0; i < count; ++i) {
        const jamie_gemm_problem& s = pr[i];
        if (s.c_panel) return jamie_fail(-1, "%s: c_panel needs a large-tile configuration [%lld %lld]", "jamie_gemm_bf16", BM, BN);
        GemmBDev& d = g.p[i];
        d.A = (const unsigned short*)s.A; d.B = (const unsigned short*)s.B; d.C = s.C; d.bias = s.bias;
        d.aux0 = s.aux0; d.partial = s.partial; d.slab_stride = s.slab_stride;
        d.M = s.M; d.N = s.N; d.K = s.K; d.lda = s.lda; d.ldb = s.ldb; d.ldc = s.ldc; d.aux_ld = s.aux_ld;
        d.splitk = s.splitk < 1 ? 1 : s.splitk;
        int kc = (s.K + d.splitk - 1) / d.splitk;
        kc = ((kc + BK - 1) / BK) * BK;
        d.kchunk = kc;
        d.tiles_m = (s.M + BM - 1) / BM;
        d.tiles_n = (s.N + BN - 1) / BN;
        d.n_tiles = d.tiles_m * d.tiles_n * d.splitk;
        g.ntiles[i] = d.n_tiles;
        tiles += d.n_tiles;
        d.epi = s.epi; d.accumulate = s.accumulate; d.scale = s.scale; d.pscale = s.pscale;
        if (s.b_tr || s.a_tr || s.c_bf16) return jamie_fail(-1, "%s: a_tr / b_tr / c_bf16 need a large-tile LDS-DMA configuration [%lld %lld]", "jamie_gemm_bf16", BM, BN);
        if (s.partial && s.epi == JAMIE_EPI_STORE)
            return jamie_fail(-1, "%s: sum-of-squares partials of a plain store need a large-tile configuration [%lld %lld]", "jamie_gemm_bf16", BM, BN);
        d.a_bytes = (unsigned)(((long long)(s.M - 1) * s.lda + s.K) * 2);
        d.b_bytes = (unsigned)(((long long)(s.N - 1) * s.ldb + s.K) * 2);
        if (s.M <= 64 || s.N <= 64 || s.K <= 64) big = false;
    }
    if (tiles == 0) return 0;
    if (big)
        hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, BK, WM, WN, 1, D>), dim3(tiles), dim3(WM * WN * 64), 0, st, g);
    else
        hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, BK, WM, WN, 0, D>), dim3(tiles), dim3(WM * WN * 64), 0, st, g);
    return jamie_launch_status("jamie_gemm_bf16");
}

template <int BM, int BN, int WM, int WN, int NB, int ILV = 0, int V2 = 0>
static int launch_dma(const jamie_gemm_problem* pr, int count, hipStream_t st, const RangeRide* rr = nullptr, int ride_blocks = 0) {
    constexpr int BK = 64;
    GemmBGroup g;
    memset(&g, 0, sizeof(g));
    g.count = count;
    int tiles = 0;
    bool big = true;
    for (int i = 0; i < count; ++i) {
        const jamie_gemm_problem& s = pr[i];
        GemmBDev& d = g.p[i];
        d.A = (const unsigned short*)s.A; d.B = (const unsigned short*)s.B; d.C = s.C; d.bias = s.bias;
        d.aux0 = s.aux0; d.partial = s.partial; d.slab_stride = s.slab_stride;
        d.M = s.M; d.N = s.N; d.K = s.K; d.lda = s.lda; d.ldb = s.ldb; d.ldc = s.ldc; d.aux_ld = s.aux_ld;
        d.splitk = s.splitk < 1 ? 1 : s.splitk;
        int kc = (s.K + d.splitk - 1) / d.splitk;
        kc = ((kc + BK - 1) / BK) * BK;
        d.kchunk = kc;
        d.tiles_m = (s.M + BM - 1) / BM;
        d.tiles_n = (s.N + BN - 1) / BN;
        d.n_tiles = d.tiles_m * d.tiles_n * d.splitk;
        g.ntiles[i] = d.n_tiles;
        tiles += d.n_tiles;
        d.epi = s.epi; d.accumulate = s.accumulate; d.scale = s.scale; d.pscale = s.pscale;
        d.b_tr = s.b_tr; d.a_tr = s.a_tr; d.store_nt = s.store_nt; d.c_bf16 = s.c_bf16; d.c_panel = s.c_panel;
        if (s.c_panel && !(V2 && s.epi == JAMIE_EPI_STORE && !s.accumulate && !s.c_bf16 && s.N % 4 == 0 && (uintptr_t)s.C % 16 == 0 &&
                           s.slab_stride % 4 == 0 && (!s.bias || (uintptr_t)s.bias % 16 == 0) &&
                           (d.splitk == 1 || s.slab_stride >= (long long)((s.N + JAMIE_PANEL - 1) / JAMIE_PANEL) * JAMIE_PANEL * s.M) &&
                           (long long)((s.N + JAMIE_PANEL - 1) / JAMIE_PANEL) * s.M * JAMIE_PANEL * 4 < 0xFFFFFFF0LL))
            return jamie_fail(-1, "%s: c_panel needs a large-tile configuration, a plain fp32 store, N %% 4 == 0, 16-byte aligned C / bias, "
                                  "slab_stride >= ceil(N / JAMIE_PANEL) * JAMIE_PANEL * M [%lld %lld]", "jamie_gemm_bf16", BM, BN);
        if (s.c_bf16 && !(V2 && s.epi == JAMIE_EPI_STORE && !s.accumulate && d.splitk == 1 && s.bias == nullptr))
            return jamie_fail(-1, "%s: c_bf16 needs a large-tile configuration, a plain store, no bias, no accumulate, splitk == 1 [%lld %lld]",
                              "jamie_gemm_bf16", BM, BN);
        if (s.a_tr && !(V2 && BM == 128 && BN == 128 && s.b_tr))
            return jamie_fail(-1, "%s: a_tr (A stored [K, M]) needs b_tr and a 128 x 128 large-tile configuration (24, 25, 29, 30) [%lld %lld]",
                              "jamie_gemm_bf16", BM, BN);
        if (s.partial && s.epi == JAMIE_EPI_STORE && (!V2 || d.splitk != 1))
            return jamie_fail(-1, "%s: sum-of-squares partials of a plain store need a large-tile configuration and splitk == 1 [%lld %lld]",
                              "jamie_gemm_bf16", BM, BN);
        if (s.b_tr && !(V2 && BN == 128))
            return jamie_fail(-1, "%s: b_tr (B stored [K, N]) needs a large-tile configuration with 128 columns (23, 24, 25) [%lld %lld]",
                              "jamie_gemm_bf16", BM, BN);
        if (s.M <= 64 || s.N <= 64 || s.K <= 64) big = false;
    }
    if (tiles == 0) return 0;
    if constexpr (V2) {
        for (int i = 0; i < count; ++i) {
            const jamie_gemm_problem& s = pr[i];
            g.p[i].vec = (s.c_panel || s.ldc % 4 == 0) && ((uintptr_t)s.C % (s.c_bf16 ? 8 : 16) == 0) && (s.slab_stride % 4 == 0) &&
                         (!s.bias || (uintptr_t)s.bias % 16 == 0) &&
                         (s.epi != JAMIE_EPI_MSE || (s.aux_ld % 4 == 0 && (uintptr_t)s.aux0 % 16 == 0));
        }
        bool trm = false;
        for (int i = 0; i < count; ++i) trm = trm || pr[i].a_tr || pr[i].b_tr;
        if constexpr (BM == 128 && BN == 128 && WM == 2 && WN == 4 && NB == 2) {
            if (rr) {       // (jamie_gemm_bf16_ranges: configuration 29 only)
                if (big)
                    hipLaunchKernelGGL((gemm_bf16_dma2_ride_kernel<BM, BN, WM, WN, 1, NB, 1>), dim3(tiles + ride_blocks), dim3(WM * WN * 64), 0, st, g, *rr, tiles);
                else
                    hipLaunchKernelGGL((gemm_bf16_dma2_ride_kernel<BM, BN, WM, WN, 0, NB, 1>), dim3(tiles + ride_blocks), dim3(WM * WN * 64), 0, st, g, *rr, tiles);
                return jamie_launch_status("jamie_gemm_bf16_ranges");
            }
        }
        if (rr) return jamie_fail(-1, "%s: riders need tile configuration 29 [%lld %lld]", "jamie_gemm_bf16_ranges", BM, BN);
        if constexpr (BN == 128) {
            if (trm) {
                if (big)
                    hipLaunchKernelGGL((gemm_bf16_dma2_kernel<BM, BN, WM, WN, 1, NB, 1>), dim3(tiles), dim3(WM * WN * 64), 0, st, g);
                else
                    hipLaunchKernelGGL((gemm_bf16_dma2_kernel<BM, BN, WM, WN, 0, NB, 1>), dim3(tiles), dim3(WM * WN * 64), 0, st, g);
                return jamie_launch_status("jamie_gemm_bf16");
            }
        }
        if (big)
            hipLaunchKernelGGL((gemm_bf16_dma2_kernel<BM, BN, WM, WN, 1, NB, 0>), dim3(tiles), dim3(WM * WN * 64), 0, st, g);
        else
            hipLaunchKernelGGL((gemm_bf16_dma2_kernel<BM, BN, WM, WN, 0, NB, 0>), dim3(tiles), dim3(WM * WN * 64), 0, st, g);
        return jamie_launch_status("jamie_gemm_bf16");
    }
    if (rr) return jamie_fail(-1, "%s: riders need tile configuration 29 [%lld %lld]", "jamie_gemm_bf16_ranges", BM, BN);
    if (big)
        hipLaunchKernelGGL((gemm_bf16_dma_kernel<BM, BN, WM, WN, 1, NB, ILV>), dim3(tiles), dim3(WM * WN * 64), 0, st, g);
    else
        hipLaunchKernelGGL((gemm_bf16_dma_kernel<BM, BN, WM, WN, 0, NB, ILV>), dim3(tiles), dim3(WM * WN * 64), 0, st, g);
    return jamie_launch_status("jamie_gemm_bf16");
}

#ifdef JAMIE_EXPERIMENTS
template <int BM, int BN, int WM, int WN, int NB>
static int launch_dma_bn(const jamie_gemm_problem* pr, const jamie_bnact_fwd_problem* bn, int count, float p_drop, float momentum,
                         float eps, float slope, const uint64_t* rng, unsigned* tickets, int n_tickets, int mode, hipStream_t st) {
    constexpr int BK = 64;
    GemmBGroup g;
    BnFuse f;
    memset(&g, 0, sizeof(g));
    memset(&f, 0, sizeof(f));
    g.count = count;
    int tiles = 0, strips = 0;
    bool need_rng = false;
    for (int i = 0; i < count; ++i) {
        const jamie_gemm_problem& s = pr[i];
        const jamie_bnact_fwd_problem& b = bn[i];
        GemmBDev& d = g.p[i];
        d.A = (const unsigned short*)s.A; d.B = (const unsigned short*)s.B; d.C = s.C; d.bias = s.bias;
        d.slab_stride = s.slab_stride;
        d.M = s.M; d.N = s.N; d.K = s.K; d.lda = s.lda; d.ldb = s.ldb; d.ldc = s.ldc;
        d.splitk = s.splitk < 1 ? 1 : s.splitk;
        int kc = (s.K + d.splitk - 1) / d.splitk;
        d.kchunk = ((kc + BK - 1) / BK) * BK;
        d.tiles_m = (s.M + BM - 1) / BM;
        d.tiles_n = (s.N + BN - 1) / BN;
        d.n_tiles = d.tiles_m * d.tiles_n * d.splitk;
        g.ntiles[i] = d.n_tiles;
        tiles += d.n_tiles;
        d.epi = JAMIE_EPI_STORE; d.scale = 1.f; d.pscale = 1.f; d.vec = 1;
        JAMIE_ARG(s.epi == JAMIE_EPI_STORE && !s.accumulate && !s.a_tr && !s.b_tr && !s.c_bf16 && !s.partial,
                  "fused BatchNorm: a plain forward product (K-contiguous operands, fp32 slabs, no partial sums)");
        JAMIE_ARG(s.M <= BN4_MAXR * BN4_RP && s.N % 4 == 0 && s.ldc == s.N && (uintptr_t)s.C % 16 == 0 && s.slab_stride % 4 == 0 &&
                      (!s.bias || (uintptr_t)s.bias % 16 == 0),
                  "fused BatchNorm: batch <= 512, N a multiple of 4, dense 16-byte aligned slabs");
        JAMIE_ARG((long long)s.M * s.ldc * 4 < 0x7FFFFFF0LL, "fused BatchNorm: a slab must stay below 2 GiB");
        JAMIE_ARG(b.h == s.C && b.B == s.M && b.N == s.N && b.nslab == d.splitk &&
                      (d.splitk == 1 || b.slab_stride == s.slab_stride),
                  "fused BatchNorm: the BatchNorm problem must describe the product's own slab buffer");
        JAMIE_ARG(b.gamma && b.beta && b.running_mean && b.running_var && b.save_mean && b.save_invstd, "null pointer");
        JAMIE_ARG((b.out || b.out_bf16) && !b.outT_bf16 && (uintptr_t)b.out % 16 == 0 && (uintptr_t)b.out_bf16 % 8 == 0 &&
                      (uintptr_t)b.mask % 4 == 0,
                  "fused BatchNorm: fp32 and / or row-major bf16 output, aligned; no transposed copy");
        JAMIE_ARG(((long long)(b.nslab - 1) * b.slab_stride + (long long)b.B * b.N) * 4 < 0xFFFFFFF0LL,
                  "activation slabs must stay below 4 GiB");
        BnFwdDev& q = f.p[i];
        q.h = b.h; q.gamma = b.gamma; q.beta = b.beta; q.rmean = b.running_mean; q.rvar = b.running_var;
        q.smean = b.save_mean; q.sinvstd = b.save_invstd; q.out = b.out; q.mask = b.mask;
        q.out_bf = (unsigned short*)b.out_bf16; q.outT_bf = nullptr;
        q.slab_stride = b.slab_stride; q.nslab = b.nslab; q.B = b.B; q.N = b.N; q.rng_stream = b.rng_stream;
        q.panel = 0;
        f.ticket_base[i] = strips;
        strips += d.tiles_n;
        if (!b.mask && p_drop > 0.f) need_rng = true;
        d.a_bytes = (unsigned)(((long long)(s.M - 1) * s.lda + s.K) * 2);
        d.b_bytes = (unsigned)(((long long)(s.N - 1) * s.ldb + s.K) * 2);
    }
    JAMIE_ARG(!need_rng || rng != nullptr, "rng state required when no explicit mask is given");
    JAMIE_ARG(tickets != nullptr && n_tickets >= 4 + 2 * strips, "tickets: 4 + 2 words per 128-column strip, zero-initialised");
    if (tiles == 0) return 0;
    f.tickets = tickets; f.rng = rng; f.p_drop = p_drop; f.momentum = momentum; f.eps = eps; f.slope = slope; f.mode = mode;
    hipLaunchKernelGGL((gemm_bf16_dma2_bn_kernel<BM, BN, WM, WN, 1, NB>), dim3(tiles), dim3(WM * WN * 64), 0, st, g, f);
    return jamie_launch_status("jamie_gemm_bf16_bn");
}
#endif

static const int BT[35][2] = {{128, 128}, {64, 64}, {64, 64}, {32, 64}, {64, 64}, {64, 64}, {64, 64}, {64, 64}, {64, 64}, {128, 64}, {64, 64}, {128, 64},
                              {128, 128}, {128, 128}, {128, 128}, {256, 128}, {128, 128}, {128, 128},
                              {64, 64}, {128, 128}, {256, 128}, {256, 128}, {64, 64},
                              {256, 128}, {128, 128}, {128, 128}, {128, 256}, {64, 64}, {256, 256}, {128, 128}, {128, 128}, {256, 128}, {128, 128}, {128, 128}, {128, 128}};

// measured on the config-2 layer shapes (tools/bench_gemm_bf16.py): 64x64x64 (28 us per grouped launch) beats 128x128x64
// (41 us) at M = 512 / K = 512; the large tile only wins on large squares (742 vs 488 TFLOP/s at 4096^3)
// ... and the LDS-DMA variants beat the register-staged one: 3 LDS buffers (cfg 7) when every problem has a long K
// (forward / dX: 36 vs 40 us cold), 2 buffers (cfg 10) when K is the batch (dW, K = 512: 8 k-steps only)
static int pick_cfg_b(int max_m, int max_n, int min_k) { (void)max_m; (void)max_n; return min_k >= 1000 ? 7 : 10; }

static int gemm_bf16_impl(const jamie_gemm_problem* pr, int count, int cfg, void* stream, const RangeRide* rr, int ride_blocks) {
    JAMIE_ARG(pr != nullptr && count >= 1 && count <= JAMIE_MAX_GEMM_GROUP, "1 <= count <= JAMIE_MAX_GEMM_GROUP");
    int max_m = 0, max_n = 0, min_k = 1 << 30;
    for (int i = 0; i < count; ++i) {
        const jamie_gemm_problem& s = pr[i];
        JAMIE_ARG(s.A && s.B && s.C, "null operand");
        JAMIE_ARG(s.M > 0 && s.N > 0 && s.K > 0, "empty problem");
        JAMIE_ARG(s.ldc >= s.N && s.lda >= (s.a_tr ? s.M : s.K) && s.ldb >= (s.b_tr ? s.N : s.K), "leading dimensions");
        JAMIE_ARG(!s.a_tr || (s.M % 8 == 0 && s.M >= 8), "a_tr: M must be a multiple of 8");
        if (s.K < min_k) min_k = s.K;
        JAMIE_ARG(s.K % 8 == 0 && s.lda % 8 == 0 && s.ldb % 8 == 0, "bf16 operands need K, lda, ldb multiples of 8");
        JAMIE_ARG(!s.b_tr || (s.N % 8 == 0 && s.N >= 8 && s.epi == JAMIE_EPI_STORE), "b_tr: N must be a multiple of 8, plain store epilogue");
        JAMIE_ARG(((uintptr_t)s.A % 16) == 0 && ((uintptr_t)s.B % 16) == 0, "bf16 operands must be 16-byte aligned");
        JAMIE_ARG((s.a_tr ? ((long long)(s.K - 1) * s.lda + s.M) : ((long long)(s.M - 1) * s.lda + s.K)) * 2 < 0xFFFFFFF0LL &&
                      (s.b_tr ? ((long long)(s.K - 1) * s.ldb + s.N) : ((long long)(s.N - 1) * s.ldb + s.K)) * 2 < 0xFFFFFFF0LL,
                  "operands must stay below 4 GiB");
        JAMIE_ARG(s.epi == JAMIE_EPI_STORE || s.epi == JAMIE_EPI_MSE, "bf16 GEMM epilogues: STORE, MSE");
        JAMIE_ARG(s.epi == JAMIE_EPI_STORE || s.splitk <= 1, "fused epilogues need splitk == 1");
        JAMIE_ARG(s.splitk <= 1 || !s.accumulate, "split-K slabs cannot accumulate");
        JAMIE_ARG(s.epi != JAMIE_EPI_MSE || (s.aux0 && s.aux_ld >= s.N), "MSE epilogue needs aux0 = X");
        JAMIE_ARG(s.splitk <= 1 || s.slab_stride >= (long long)s.M * s.ldc, "slab_stride too small");
        JAMIE_ARG(s.a_rows == nullptr, "row gather is not supported in the bf16 GEMM");
        if (s.M > max_m) max_m = s.M;
        if (s.N > max_n) max_n = s.N;
    }
    hipStream_t st = (hipStream_t)stream;
    if (cfg < 0) cfg = pick_cfg_b(max_m, max_n, min_k);
    if (rr && cfg != 29) return jamie_fail(-1, "%s: riders need tile configuration 29 [%lld %lld]", "jamie_gemm_bf16_ranges", cfg, 0);
    switch (cfg) {
        case 0: return launch_b<128, 128, 64, 2, 2, 2>(pr, count, st);
        case 1: return launch_b<64, 64, 64, 2, 2, 3>(pr, count, st);
        case 2: return launch_b<64, 64, 32, 2, 2, 2>(pr, count, st);
        case 3: return launch_b<32, 64, 64, 1, 2, 2>(pr, count, st);
        case 4: return launch_b<64, 64, 64, 2, 2, 1>(pr, count, st);
        case 5: return launch_b<64, 64, 64, 2, 2, 2>(pr, count, st);
        case 6: return launch_b<64, 64, 64, 2, 2, 4>(pr, count, st);
        case 7: return launch_dma<64, 64, 2, 2, 3>(pr, count, st);
        case 8: return launch_dma<64, 64, 2, 2, 4>(pr, count, st);
        case 9: return launch_dma<128, 64, 2, 2, 3>(pr, count, st);
        case 10: return launch_dma<64, 64, 2, 2, 2>(pr, count, st);
        case 11: return launch_dma<128, 64, 2, 2, 4>(pr, count, st);
        case 12: return launch_dma<128, 128, 2, 2, 2>(pr, count, st);
        case 13: return launch_dma<128, 128, 2, 2, 3>(pr, count, st);
        case 14: return launch_dma<128, 128, 2, 4, 3>(pr, count, st);
        case 15: return launch_dma<256, 128, 2, 4, 2>(pr, count, st);
        case 16: return launch_dma<128, 128, 2, 4, 4>(pr, count, st);
        case 17: return launch_dma<128, 128, 2, 4, 5>(pr, count, st);
        case 18: return launch_dma<64, 64, 2, 2, 3, 1>(pr, count, st);
        case 19: return launch_dma<128, 128, 2, 4, 3, 1>(pr, count, st);
        case 20: return launch_dma<256, 128, 4, 2, 3>(pr, count, st);
        case 21: return launch_dma<256, 128, 4, 2, 3, 1>(pr, count, st);
        case 22: return launch_dma<64, 64, 2, 2, 2, 1>(pr, count, st);
        case 23: return launch_dma<256, 128, 4, 2, 3, 0, 1>(pr, count, st);
        case 24: return launch_dma<128, 128, 2, 2, 3, 0, 1>(pr, count, st);
        case 25: return launch_dma<128, 128, 2, 2, 2, 0, 1>(pr, count, st);
        case 26: return launch_dma<128, 256, 2, 4, 3, 0, 1>(pr, count, st);
        case 27: return launch_dma<64, 64, 2, 2, 3, 0, 1>(pr, count, st);
        case 28: return launch_dma<256, 256, 4, 4, 2, 0, 1>(pr, count, st);
        case 29: return launch_dma<128, 128, 2, 4, 2, 0, 1>(pr, count, st, rr, ride_blocks);      // 8 waves of 64x32, 2 buffers
        case 30: return launch_dma<128, 128, 4, 2, 2, 0, 1>(pr, count, st);      // 8 waves of 32x64, 2 buffers
        case 31: return launch_dma<256, 128, 4, 4, 3, 0, 1>(pr, count, st);      // 16 waves of 64x32, 3 buffers
        case 32: return launch_dma<128, 128, 2, 4, 3, 0, 1>(pr, count, st);      // 8 waves of 64x32, 3 buffers
        case 33: return launch_dma<128, 128, 2, 4, 4, 0, 1>(pr, count, st);      // ... 4 buffers (three k-tiles in flight)
        case 34: return launch_dma<128, 128, 2, 4, 5, 0, 1>(pr, count, st);      // ... 5 buffers = all 160 KB (four in flight)
        default: return jamie_fail(-1, "%s: unknown tile configuration [%lld %lld]", "jamie_gemm_bf16", cfg, 0);
    }
}

extern "C" int jamie_gemm_bf16(const jamie_gemm_problem* pr, int count, int cfg, void* stream) {
    return gemm_bf16_impl(pr, count, cfg, stream, nullptr, 0);
}

extern "C" int jamie_gemm_bf16_ranges(const jamie_gemm_problem* pr, int count, int cfg, const float* g, void* g_bf16,
                                      const long long* offsets, const long long* lengths, int n_ranges, float* partials,
                                      int n_partials, uint64_t* state, const jamie_latent_m* fin, void* stream) {
    RangeRide rr;
    int blocks = 0;
    const int rc = jamie_range_ride_fill(g, g_bf16, offsets, lengths, n_ranges, partials, n_partials, state, fin, &rr, &blocks);
    if (rc) return rc;
    return gemm_bf16_impl(pr, count, cfg, stream, &rr, blocks);
}

#ifdef JAMIE_EXPERIMENTS
extern "C" int jamie_gemm_bf16_bn(const jamie_gemm_problem* pr, const jamie_bnact_fwd_problem* bn, int count, int cfg,
                                  float p_drop, float momentum, float eps, float slope, const uint64_t* rng, unsigned* tickets,
                                  int n_tickets, int mode, void* stream) {
    JAMIE_ARG(pr != nullptr && bn != nullptr && count >= 1 && count <= JB_FUSE_MAX, "1 <= count <= 4");
    JAMIE_ARG(p_drop >= 0.f && p_drop < 1.f, "0 <= p < 1");
    JAMIE_ARG(mode == 1 || mode == 2, "mode: 1 (last arriver reduces) or 2 (every slice takes a share)");
    for (int i = 0; i < count; ++i) {
        const jamie_gemm_problem& s = pr[i];
        JAMIE_ARG(s.A && s.B && s.C, "null operand");
        JAMIE_ARG(s.M > 0 && s.N > 0 && s.K > 0, "empty problem");
        JAMIE_ARG(s.ldc >= s.N && s.lda >= s.K && s.ldb >= s.K, "leading dimensions");
        JAMIE_ARG(s.K % 8 == 0 && s.lda % 8 == 0 && s.ldb % 8 == 0, "bf16 operands need K, lda, ldb multiples of 8");
        JAMIE_ARG(((uintptr_t)s.A % 16) == 0 && ((uintptr_t)s.B % 16) == 0, "bf16 operands must be 16-byte aligned");
        JAMIE_ARG(((long long)(s.M - 1) * s.lda + s.K) * 2 < 0xFFFFFFF0LL && ((long long)(s.N - 1) * s.ldb + s.K) * 2 < 0xFFFFFFF0LL,
                  "operands must stay below 4 GiB");
        JAMIE_ARG(s.splitk <= 1 || s.slab_stride >= (long long)s.M * s.ldc, "slab_stride too small");
        JAMIE_ARG(s.a_rows == nullptr, "row gather is not supported in the bf16 GEMM");
    }
    hipStream_t st = (hipStream_t)stream;
    switch (cfg) {
        case 31: return launch_dma_bn<256, 128, 4, 4, 3>(pr, bn, count, p_drop, momentum, eps, slope, rng, tickets, n_tickets, mode, st);
        case 32: return launch_dma_bn<128, 128, 2, 4, 3>(pr, bn, count, p_drop, momentum, eps, slope, rng, tickets, n_tickets, mode, st);
        default: return jamie_fail(-1, "%s: tile configuration 31 or 32 [%lld %lld]", "jamie_gemm_bf16_bn", cfg, 0);
    }
}
#endif

extern "C" int jamie_gemm_bf16_tile(int max_m, int max_n, int cfg, int* bm, int* bn) {
    if (cfg < 0) cfg = pick_cfg_b(max_m, max_n, 1 << 30);
    if (cfg > 34 || !bm || !bn) return jamie_fail(-1, "%s: bad arguments [%lld %lld]", "jamie_gemm_bf16_tile", cfg, 0);
    *bm = BT[cfg][0]; *bn = BT[cfg][1];
    return 0;
}

// ------------------------------------------------------------------------------------------------
// fp32 [R, C] (sum of slabs) -> bf16 [R, C] and / or bf16 transposed [C, R]: 64x64 tiles through LDS so that both
// the reads and the transposed writes move whole 128-byte lines.  Up to 16 matrices per launch (the weights).
// ------------------------------------------------------------------------------------------------
#include "cast_tile.h"
__global__ __launch_bounds__(256) void cast_transpose_kernel(CastGroup g) {
    __shared__ float tile[64][65];
    cast_tile_block(g, (int)blockIdx.x, tile);
}

int jamie_cast_fill_group(const jamie_cast_problem* pr, int count, CastGroup* gp, int* nblocks) {
    JAMIE_ARG(pr && count >= 1 && count <= CT_MAX, "1 <= count <= 16");
    CastGroup& g = *gp;
    memset(&g, 0, sizeof(g));
    g.count = count;
    int blocks = 0;
    for (int i = 0; i < count; ++i) {
        const jamie_cast_problem& s = pr[i];
        JAMIE_ARG((s.src || s.src_bf16) && (s.dst || s.dstT || s.dst32) && s.R > 0 && s.C > 0 && s.ld >= s.C && s.nslab >= 1, "bad cast problem");
        JAMIE_ARG(!s.rows || s.src, "row gather needs an fp32 source");
        JAMIE_ARG(!s.dst32 || s.ld32 >= s.C, "ld32 < C");
        JAMIE_ARG(!s.src_bf16 || (!s.src && s.nslab == 1), "src_bf16 excludes src / slabs");
        JAMIE_ARG(!s.dst || s.ldd >= s.C, "ldd < C");
        JAMIE_ARG(!s.dstT || s.ldt >= s.R, "ldt < R");
        CastDev& d = g.p[i];
        d.src = s.src; d.src_bf = (const unsigned short*)s.src_bf16; d.dst = (unsigned short*)s.dst; d.dstT = (unsigned short*)s.dstT; d.slab_stride = s.slab_stride;
        d.R = s.R; d.C = s.C; d.ld = s.ld; d.ldd = s.ldd; d.ldt = s.ldt; d.nslab = s.nslab;
        d.rows = s.rows; d.dst32 = s.dst32; d.ld32 = s.ld32;
        d.blk_begin = blocks; d.tiles_c = (s.C + 63) / 64;
        blocks += ((s.R + 63) / 64) * d.tiles_c;
    }
    *nblocks = blocks;
    return 0;
}

extern "C" int jamie_cast_transpose(const jamie_cast_problem* pr, int count, void* stream) {
    CastGroup g;
    int blocks = 0;
    const int rc = jamie_cast_fill_group(pr, count, &g, &blocks);
    if (rc) return rc;
    hipLaunchKernelGGL(cast_transpose_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g);
    return jamie_launch_status("jamie_cast_transpose");
}

// ------------------------------------------------------------------------------------------------
// Reconstruction loss behind a split-K x_hat GEMM (jamie.py:637-641): y = sum of the slabs (bias already in slab 0),
// d = y - x, partial[tile] = pscale * sum d^2 per 64x64 tile (same tiling / partial count as the fused MSE epilogue
// of the 64x64 GEMM), and d * scale written as fp32 [R, C] plus the bf16 / bf16-transposed copies the backward
// products read.  Same 64x64 LDS-transposing layout as cast_transpose_kernel.
// ------------------------------------------------------------------------------------------------
// once-read loads of the MSE launch (the x_hat slabs, the batch) non-temporal: -2.5 us per step (profiles/r03_ab_more_nt.log;
// default policy: +2.5)
__device__ __forceinline__ float4 JB_MSE_LD4(const float* p) {
    return make_float4(__builtin_nontemporal_load(p), __builtin_nontemporal_load(p + 1), __builtin_nontemporal_load(p + 2),
                       __builtin_nontemporal_load(p + 3));
}
struct MseDev { const float* y; const float* x; float* d; unsigned short* dst; unsigned short* dstT; float* partial; float* colpart;
                long long slab_stride; int R, C, nslab, blk_begin, tiles_c; float scale, pscale; };
struct MseGroup { MseDev p[JAMIE_MAX_GROUP]; int count; };

// CP: per-tile column sums of d (colpart); HAS_D: the fp32 output exists.  Template constants, not run-time tests: with `if (P.d)` /
// `if (P.colpart)` inside the four passes the launch took 14.6 instead of 9.8 us in the step (profiles/r05_launch_timeline_*.txt)
template <bool CP, bool HAS_D>
__global__ __launch_bounds__(256) void mse_cast_kernel(MseGroup g) {
    __shared__ float tile[64][65];
    __shared__ float red[4];
    __shared__ float csum[4][64];
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
    int pi = 0;
    for (int i = 1; i < JAMIE_MAX_GROUP; ++i)
        if (i < g.count && (int)blockIdx.x >= g.p[i].blk_begin) pi = i;
    const MseDev& P = g.p[pi];
    const int b = blockIdx.x - P.blk_begin;
    const int r0 = (b % ((P.R + 63) / 64)) * 64, c0 = (b / ((P.R + 63) / 64)) * 64;   // tile order of the GEMM: m fastest
    const int q = threadIdx.x & 15, rr0 = threadIdx.x >> 4;
    const bool vec = (P.C % 4 == 0) && (P.slab_stride % 4 == 0);
    float local = 0.f;
    // every load of the four passes is issued before the first store (a store between two passes' loads serialises their
    // round trips: the compiler must assume the pointers alias; slabs four at a time)
    float4 yv[4], xv[4];
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        yv[pass] = xv[pass] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int r = r0 + rr0 + 16 * pass, c = c0 + 4 * q;
        if (vec && r < P.R && c < P.C) xv[pass] = JB_MSE_LD4(P.x + (long long)r * P.C + c);
    }
    if (vec) {
        for (int s0 = 0; s0 < P.nslab; s0 += 4) {
            float4 t[4][4];
#pragma unroll
            for (int pass = 0; pass < 4; ++pass) {
                const int r = r0 + rr0 + 16 * pass, c = c0 + 4 * q;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    t[pass][u] = (r < P.R && c < P.C && s0 + u < P.nslab)
                                     ? JB_MSE_LD4(P.y + (s0 + u) * P.slab_stride + (long long)r * P.C + c)
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int pass = 0; pass < 4; ++pass)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    yv[pass].x += t[pass][u].x; yv[pass].y += t[pass][u].y; yv[pass].z += t[pass][u].z; yv[pass].w += t[pass][u].w;
                }
        }
    }
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int rr = rr0 + 16 * pass, r = r0 + rr, c = c0 + 4 * q;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (r < P.R && c < P.C) {
            const long long o = (long long)r * P.C + c;
            if (vec) {
                v[0] = yv[pass].x - xv[pass].x; v[1] = yv[pass].y - xv[pass].y; v[2] = yv[pass].z - xv[pass].z; v[3] = yv[pass].w - xv[pass].w;
                local += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
                v[0] *= P.scale; v[1] *= P.scale; v[2] *= P.scale; v[3] *= P.scale;
                if constexpr (HAS_D) *reinterpret_cast<float4*>(P.d + o) = make_float4(v[0], v[1], v[2], v[3]);
                if (P.dst) {
                    const unsigned short b0 = __builtin_bit_cast(unsigned short, (__bf16)v[0]), b1 = __builtin_bit_cast(unsigned short, (__bf16)v[1]);
                    const unsigned short b2 = __builtin_bit_cast(unsigned short, (__bf16)v[2]), b3 = __builtin_bit_cast(unsigned short, (__bf16)v[3]);
                    *reinterpret_cast<uint2*>(P.dst + o) = make_uint2((unsigned)b0 | ((unsigned)b1 << 16), (unsigned)b2 | ((unsigned)b3 << 16));
                }
            } else {
                for (int e = 0; e < 4; ++e) {
                    if (c + e >= P.C) continue;
                    for (int s = 0; s < P.nslab; ++s) v[e] += P.y[s * P.slab_stride + o + e];
                    v[e] -= P.x[o + e];
                    local += v[e] * v[e];
                    v[e] *= P.scale;
                    if constexpr (HAS_D) P.d[o + e] = v[e];
                    if (P.dst) P.dst[o + e] = __builtin_bit_cast(unsigned short, (__bf16)v[e]);
                }
            }
        }
        tile[rr][4 * q] = v[0]; tile[rr][4 * q + 1] = v[1]; tile[rr][4 * q + 2] = v[2]; tile[rr][4 * q + 3] = v[3];
        if constexpr (CP) { cs[0] += v[0]; cs[1] += v[1]; cs[2] += v[2]; cs[3] += v[3]; }    // (column sums of d: this thread's rows rr0 + 16 pass)
    }
    const float tot = block_sum(local, red);
    if (threadIdx.x == 0 && P.partial) P.partial[b] = tot * P.pscale;
    if constexpr (CP) {         // column sums of this tile's 64 rows of d (rows beyond R hold 0), in a fixed order:
        // a thread's four rows (rr0 + 16 pass), the four row phases of its wave by xor-shuffles, the four waves through LDS
        // (a serial 64-row loop by 64 threads cost the launch 1.7 us: profiles/r05_ab_mse_colpart.log)
#pragma unroll
        for (int e = 0; e < 4; ++e) { cs[e] += __shfl_xor(cs[e], 16); cs[e] += __shfl_xor(cs[e], 32); }
        const int wvx = threadIdx.x >> 6;
        if ((threadIdx.x & 63) < 16) {
#pragma unroll
            for (int e = 0; e < 4; ++e) csum[wvx][4 * q + e] = cs[e];
        }
        __syncthreads();
        const int cc = threadIdx.x;
        if (cc < 64 && c0 + cc < P.C) P.colpart[(long long)(r0 >> 6) * P.C + c0 + cc] = (csum[0][cc] + csum[1][cc]) + (csum[2][cc] + csum[3][cc]);
    }
    if (!P.dstT) return;
    __syncthreads();
    const bool vect = (P.R % 4 == 0);
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int cc = rr0 + 16 * pass, c = c0 + cc, r = r0 + 4 * q;
        if (c >= P.C) continue;
        const unsigned short b0 = __builtin_bit_cast(unsigned short, (__bf16)tile[4 * q][cc]);
        const unsigned short b1 = __builtin_bit_cast(unsigned short, (__bf16)tile[4 * q + 1][cc]);
        const unsigned short b2 = __builtin_bit_cast(unsigned short, (__bf16)tile[4 * q + 2][cc]);
        const unsigned short b3 = __builtin_bit_cast(unsigned short, (__bf16)tile[4 * q + 3][cc]);
        unsigned short* dp = P.dstT + (long long)c * P.R + r;
        if (vect && r + 3 < P.R) {
            *reinterpret_cast<uint2*>(dp) = make_uint2((unsigned)b0 | ((unsigned)b1 << 16), (unsigned)b2 | ((unsigned)b3 << 16));
        } else {
            if (r < P.R) dp[0] = b0;
            if (r + 1 < P.R) dp[1] = b1;
            if (r + 2 < P.R) dp[2] = b2;
            if (r + 3 < P.R) dp[3] = b3;
        }
    }
}

extern "C" int jamie_mse_cast(const jamie_mse_problem* pr, int count, void* stream) {
    JAMIE_ARG(pr && count >= 1 && count <= JAMIE_MAX_GROUP, "1 <= count <= JAMIE_MAX_GROUP");
    MseGroup g;
    memset(&g, 0, sizeof(g));
    g.count = count;
    int blocks = 0;
    for (int i = 0; i < count; ++i) {
        const jamie_mse_problem& s = pr[i];
        JAMIE_ARG(s.y && s.x && s.R > 0 && s.C > 0 && s.nslab >= 1, "null pointer / empty problem");
        JAMIE_ARG(s.d || s.d_bf16 || s.dT_bf16, "no output requested");
        JAMIE_ARG(((uintptr_t)s.y % 16) == 0 && ((uintptr_t)s.x % 16) == 0 && ((uintptr_t)s.d % 16) == 0 &&
                      ((uintptr_t)s.d_bf16 % 8) == 0 && ((uintptr_t)s.dT_bf16 % 8) == 0,
                  "y, x, d must be 16-byte aligned (bf16 outputs 8-byte)");
        MseDev& d = g.p[i];
        d.y = s.y; d.x = s.x; d.d = s.d; d.dst = (unsigned short*)s.d_bf16; d.dstT = (unsigned short*)s.dT_bf16;
        d.partial = s.partial; d.slab_stride = s.slab_stride; d.R = s.R; d.C = s.C; d.nslab = s.nslab;
        d.scale = s.scale; d.pscale = s.pscale; d.colpart = s.colpart;
        d.blk_begin = blocks; d.tiles_c = (s.C + 63) / 64;
        blocks += ((s.R + 63) / 64) * d.tiles_c;
    }
    // (every problem of a launch alike: with / without the fp32 output, with / without the column sums)
    bool cp = pr[0].colpart != nullptr, has_d = pr[0].d != nullptr;
    for (int i = 1; i < count; ++i)
        JAMIE_ARG((pr[i].colpart != nullptr) == cp && (pr[i].d != nullptr) == has_d, "every problem with / without d and colpart alike");
    if (cp && has_d) hipLaunchKernelGGL((mse_cast_kernel<true, true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, g);
    else if (cp) hipLaunchKernelGGL((mse_cast_kernel<true, false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, g);
    else if (has_d) hipLaunchKernelGGL((mse_cast_kernel<false, true>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, g);
    else hipLaunchKernelGGL((mse_cast_kernel<false, false>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, g);
    return jamie_launch_status("jamie_mse_cast");
}

// fp32 GEMM on v_mfma_f32_32x32x2_f32 for JAMIE's Linear layers (forward NT, dX NN, dW TN), gfx950.
//
// Replaces the `addmm`/`mm` ATen dispatches of nn.Linear forward/backward
// (reference model.py:151,161,180,185,192,197,207; autograd at jamie.py:734).
//
// Design (MI355X_MICROARCH.md § Matrix cores; cdna_hip_programming.md §3 'FP32-input MFMA'):
//   * exact-f32 MFMA 32x32x2 runs at 64 cyc/SIMD; one accumulation chain per 32x32 tile already issues
//     back to back, so the kernel is MFMA-issue bound as long as LDS reads and global loads hide.
//   * block tile BM x BN x BK, WM x WN waves, each wave TM x TN tiles of 32x32.
//   * operands are staged global -> VGPR -> LDS in the layout they have in HBM (no transposes):
//       K-contiguous operand ("KC"): LDS [rows][BK+4]; a lane reads ONE ds_read_b128 = 4 consecutive k
//                                    feeding 4 MFMAs (row stride 16*odd bytes -> conflict-free);
//       row-contiguous operand ("RC"): LDS [BK][rows+4]; a lane reads 4 ds_read_b32 (lanes consecutive).
//     MFMA step s of a k-group of 8 consumes logical k = kk + 4*(lane>>5) + s for BOTH operands.
//   * double-buffered LDS, next tile's global loads issued before the MFMAs of the current tile.
//   * grouped launch: up to JAMIE_MAX_GEMM_GROUP_F32 problems (the modalities) share one grid; block ids are
//     remapped so that the M-tiles that re-read one weight panel run back to back on one XCD (its L2).
//   * split-K writes fp32 slabs (deterministic; the consumer kernel sums them).
//   * configurations 20 / 21 (template argument X3): the same fp32 problems on the bf16 matrix pipe -- every element cut into
//     three bf16 pieces on its way to LDS, six v_mfma_f32_32x32x16_bf16 per product, fp32-level error; the engine's default for
//     the large layers since round 5 (see the kernel's X3 branch).
#include "common.h"
#include <type_traits>

// diagnostic ablations of the register-staged k-loop (tools/ablate_gemm.sh; results are wrong, timings only), a bit mask:
// 1 = no global loads in the k-loop, 2 = no MFMAs, 4 = no LDS writes, 8 = no barrier, 16 = no fragment reads
#ifndef JAMIE_GEMM_ABL
#define JAMIE_GEMM_ABL 0
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));

// compile-time loop: f(std::integral_constant<int, I>{}) for I = BEGIN .. END-1
template <int I, int END, typename F>
__device__ __forceinline__ void jf_static_for(F&& f) {
    if constexpr (I < END) {
        f(std::integral_constant<int, I>{});
        jf_static_for<I + 1, END>(f);
    }
}

// bf16x3 k-loop, row-contiguous operands: which 4 k x 4 m block a lane stages (see the loop's note on 8-byte LDS stores).
// JAMIE_X3_HALF16 (A/B): the two 8-byte halves of a chunk 16 lanes apart -- a third of the dW launch's LDS cycles were conflicts
#ifdef JAMIE_X3_HALF16
#define JF_X3_KQ(lane) (((lane) >> 4) & 1)
#define JF_X3_C4(lane) (((lane) & 15) + 16 * ((lane) >> 5))
#else
#define JF_X3_KQ(lane) (((lane) >> 3) & 1)
#define JF_X3_C4(lane) (((lane) & 7) + 8 * ((lane) >> 4))
#endif

// diagnostic build only (tools/stamp_gemm_bf16.sh): in-kernel s_memrealtime stamps, see gemm_bf16.hip
#ifdef JAMIE_GEMMB_STAMP
__device__ unsigned long long jamie_dbg_stamps_f32[8192 * 8];
#define JF_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 8192) jamie_dbg_stamps_f32[blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define JF_STAMPV(k, v) do { if (threadIdx.x == 0 && blockIdx.x < 8192) jamie_dbg_stamps_f32[blockIdx.x * 8 + (k)] = (unsigned long long)(v); } while (0)
extern "C" int jamie_debug_stamps_f32(unsigned long long* host_out, int n_blocks) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(jamie_dbg_stamps_f32), sizeof(unsigned long long) * 8 * n_blocks);
}
// k-loop phases of waves 0, 5, 10, 15 of every workgroup at its middle k-step: [wave slot][loop top, MFMAs issued, tile stored, past barrier]
__device__ unsigned long long jamie_dbg_kstamps_f32[8192 * 16];
#define JF_KSTAMP(kt, nk, k) do { if ((kt) == (nk) / 2 && (threadIdx.x & 63) == 0 && (threadIdx.x >> 6) % 5 == 0 && blockIdx.x < 8192) \
    jamie_dbg_kstamps_f32[blockIdx.x * 16 + ((threadIdx.x >> 6) / 5) * 4 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" int jamie_debug_kstamps_f32(unsigned long long* host_out, int n_blocks) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(jamie_dbg_kstamps_f32), sizeof(unsigned long long) * 16 * n_blocks);
}
#else
#define JF_STAMP(k) do {} while (0)
#define JF_STAMPV(k, v) do {} while (0)
#define JF_KSTAMP(kt, nk, k) do {} while (0)
#endif

struct GemmDev {
    const float* A; const float* B; float* C; const float* bias;
    const float* aux0; const float* aux1; const float* aux2; const float* aux3;
    float* partial; const int32_t* a_rows;
    long long slab_stride;
    int M, N, K, lda, ldb, ldc, aux_ld;
    int splitk, kchunk, tiles_m, tiles_n, tile_begin, n_tiles;
    int epi, accumulate, a_vec, b_vec, store_nt;
    unsigned a_bytes, b_bytes;
    unsigned c_bytes;    // extent of one output slab in bytes (0: 4 GiB or more -- the generic epilogue)
    unsigned inv_tm, inv_tn;     // floor(2^32 / tiles_m) + 1, ... / tiles_n) + 1 (0: 65536 tiles or more)
    float scale, slope, eps, pscale;
};

struct GemmGroup {
    int ntiles[JAMIE_MAX_GEMM_GROUP_F32];        // tile count of every problem (0: unused), read first by every workgroup
    GemmDev p[JAMIE_MAX_GEMM_GROUP_F32];
    int count;
};

__device__ __forceinline__ float4 ld4_guard(const float* p, int nvalid, bool vec) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (nvalid >= 4 && vec) {
        v = *reinterpret_cast<const float4*>(p);
    } else {
        if (nvalid > 0) v.x = p[0];
        if (nvalid > 1) v.y = p[1];
        if (nvalid > 2) v.z = p[2];
        if (nvalid > 3) v.w = p[3];
    }
    return v;
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Branch-free staging load: a raw buffer load returns 0 for an offset beyond the descriptor's num_records, so
// invalid rows / k-slices are expressed as an out-of-range offset and the in-row tails as selects.  (The
// guarded flat-load version serialises every load behind its own branch + s_waitcnt vmcnt(0).)
#define JAMIE_OOB 0xFFFFFFF0u
__device__ __forceinline__ float4 buf_ld4(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off, unsigned scalar_off = 0) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, (int)scalar_off, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
// applied when the tile is written to LDS (AFTER the MFMAs of the previous tile), so that the loads stay
// in flight under the compute phase
__device__ __forceinline__ float4 mask4(float4 v, int nvalid) {
    float4 f;
    f.x = nvalid > 0 ? v.x : 0.f;
    f.y = nvalid > 1 ? v.y : 0.f;
    f.z = nvalid > 2 ? v.z : 0.f;
    f.w = nvalid > 3 ? v.w : 0.f;
    return f;
}

// FAST = 1 / 2 : operands addressed through buffer descriptors (needs 16-byte aligned bases, ld % 4 == 0,
//               < 4 GiB per operand, no row gather); 2: no operand row ends inside a float4 either (K % 4 == 0 for a
//               K-contiguous operand, M / N % 4 == 0 otherwise: every layer of every BASELINE configuration) -- the LDS
//               writes need no masks; FAST = 0: guarded flat loads (any shape/alignment).
// TAG = 1 marks launches in which every problem is one of the big d <-> 2d Linear products: the same code under
// a second kernel symbol, so that per-kernel profiles (rocprofv3 --stats) of the dominant GEMMs are not mixed with
// the skinny heads / latent products.
// MID = true: the k-loop with its one barrier in the MIDDLE of a k-step's MFMA stream (see the loop).
// X3 != 0: the SAME products on the bf16 matrix pipe (16x the fp32 MFMA rate).  Every fp32 operand element is cut -- once, when
// its tile goes from the staging registers to LDS -- into three bf16 pieces by truncation, x = hi + mid + lo EXACTLY (8 + 8 + 8
// significand bits; x - hi and (x - hi) - mid are exact in fp32), the LDS holds three bf16 planes per operand, and a k-block of 16
// is six v_mfma_f32_32x32x16_bf16 into the one fp32 accumulator: lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi (every bf16 x bf16
// product is exact in fp32; the three dropped terms are at most 2^-21 and typically 2^-24 of |a||b|: the rounding level of an fp32 product itself).
// Non-finite inputs come out as NaN (inf - inf in the cut), where the fp32 pipe would give inf.  See the k-loop for the LDS image.
template <int BM, int BN, int BK, int WM, int WN, bool A_KC, bool B_KC, int FAST, int TAG, bool MID = false, int X3 = 0>
__global__ __launch_bounds__(WM * WN * 64) __attribute__((amdgpu_waves_per_eu(WM * WN >= 16 && BM * BN <= 128 * 128 ? 8 : 1)))      // (16 waves of 32x32: two workgroups per CU)
void gemm_f32_kernel(GemmGroup g) {
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    static_assert(!X3 || ((BM == 128 || BM == 256) && BN == 128 && BK == 32 && NT == 256 && WM == 2 && FAST != 0 && !MID),
                  "bf16x3: 128 x 128 x 32 or 256 x 128 x 32 on four waves");
    // (row-contiguous images need no padding: a 32-lane group of ds_read_b32 reads 32 consecutive dwords of ONE k row, an 8-lane
    //  group of ds_write_b128 writes 128 contiguous bytes; rows a multiple of 256 bytes apart let all four k rows of a fragment
    //  come from one base register by ds_read2st64_b32 -- with the 4-dword pad every pair needed an address add of its own)
    constexpr int A_LD = A_KC ? BK + 4 : BM;
    constexpr int B_LD = B_KC ? BK + 4 : BN;
    constexpr int A_SZ = A_KC ? BM * A_LD : BK * A_LD;
    constexpr int B_SZ = B_KC ? BN * B_LD : BK * B_LD;
    constexpr int LA = BM * BK / 4 / NT, LB = BN * BK / 4 / NT;
    static_assert(LA >= 1 && LB >= 1 && BK % 8 == 0, "tile/thread mismatch");
    static_assert((BM * BK / 4) % NT == 0 && (BN * BK / 4) % NT == 0, "tile/thread mismatch");
    // (X3: three (BM = 128) or two (BM = 256) stages of three bf16 planes per operand, 64-byte rows of 32 k: 144 KB)
    __shared__ __attribute__((aligned(16))) float smem[X3 ? (BM == 128 ? 3 : 2) * 3 * (BM + BN) * BK / 2 : 2 * (A_SZ + B_SZ)];
    __shared__ float red[WM * WN];

    // ---- block -> (problem, tile), XCD-aware.  Hardware deals consecutive block ids round-robin over the 8
    // XCDs (block b and b+8 share an L2).  Every problem's tile list (m fastest, so the M-tiles that re-read
    // one weight panel are adjacent) is cut into 8 contiguous chunks, one per XCD, and an XCD walks its chunk
    // of problem 0, then of problem 1, ...: each XCD gets the same number of tiles OF EACH PROBLEM (the
    // modalities have different K, so mixing them unevenly left whole XCDs with only long or only short
    // tiles), and a weight panel is fetched from HBM into one L2 only.  Remainders are dealt round-robin
    // (rotation o_p) so the chunk sizes add up to exactly the number of blocks each XCD receives.
    JF_STAMP(0);
    JF_STAMPV(6, __builtin_amdgcn_s_memtime());      // shader-clock ticks (slot 7 at the end): the clock the launch ran at

    const int bid = blockIdx.x;
    const int xcd = bid & 7;
    int slot = bid >> 3;
    int pi = 0, t = 0, rot = 0;
    // (branch-free, every problem's tile count loaded up front -- unused problems hold 0: written as a loop of guarded
    //  iterations this was one dependent scalar-memory round trip and three branches per problem, 2-3 us of a tile's time on its CU
    //  slot before the first load, more beside a workgroup that saturates the matrix pipe)
    int ntl[JAMIE_MAX_GEMM_GROUP_F32];
#pragma unroll
    for (int i = 0; i < JAMIE_MAX_GEMM_GROUP_F32; ++i) ntl[i] = g.ntiles[i];
    bool found = false;
    const int n_prob = g.count;
#pragma unroll
    for (int i = 0; i < JAMIE_MAX_GEMM_GROUP_F32; ++i) {
        if (i >= n_prob) break;                    // (one scalar branch: the forward / dX launches hold 2 of 12 problems)
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
    const GemmDev& P = g.p[pi];          // (by value, as gemm_bf16.hip does: +5 us per fp32 step)
    // t -> (M tile, N tile, K slice) by multiply-high with reciprocals from the host (exact below 65536 tiles; 0: divide -- a
    // scalar integer division is ~30 instructions through the vector unit's reciprocal and back)
    int tm_i, tn_i, ks;
    if (P.inv_tm != 0) {
        const unsigned q1 = P.tiles_m == 1 ? (unsigned)t : __umulhi((unsigned)t, P.inv_tm);      // (2^32 / 1 + 1 does not fit)
        const unsigned q2 = P.tiles_n == 1 ? q1 : __umulhi(q1, P.inv_tn);
        tm_i = t - (int)q1 * P.tiles_m;
        tn_i = (int)q1 - (int)q2 * P.tiles_n;
        ks = (int)q2;
    } else {
        tm_i = t % P.tiles_m;
        tn_i = (t / P.tiles_m) % P.tiles_n;
        ks = t / (P.tiles_m * P.tiles_n);
    }
    const int m0 = tm_i * BM, n0 = tn_i * BN;
    const int kbeg = ks * P.kchunk;
    const int kend = min(P.K, kbeg + P.kchunk);
    const int nk = (kend - kbeg + BK - 1) / BK;
    JF_STAMPV(4, pi * 1000 + nk);
    JF_STAMPV(5, __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) * 1000 + __builtin_amdgcn_s_getreg(((8 - 1) << 11) | (8 << 6) | 4));

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm0 = (wid / WN) * (TM * 32), wn0 = (wid % WN) * (TN * 32);
    const int r = lane & 31, h = lane >> 5;
    const bool avec = P.a_vec != 0, bvec = P.b_vec != 0;

    // ---- per-thread staging descriptors ----
    const float* a_ptr[LA]; int a_lim[LA]; int a_lds[LA]; int a_k[LA];
    const float* b_ptr[LB]; int b_lim[LB]; int b_lds[LB]; int b_k[LB];
#pragma unroll
    for (int j = 0; j < LA; ++j) {
        const int f = tid + j * NT;
        if (A_KC) {
            const int row = f / (BK / 4), c4 = f % (BK / 4);
            const int gm = m0 + row;
            const bool ok = gm < P.M;
            const long long grow = ok ? (P.a_rows ? (long long)P.a_rows[gm] : (long long)gm) : 0;
            a_ptr[j] = P.A + grow * P.lda + c4 * 4;
            a_lim[j] = ok ? 1 : 0;      // row valid
            a_k[j] = c4 * 4;            // k offset within tile
            a_lds[j] = row * A_LD + c4 * 4;
        } else {
            // (X3: a thread takes a 4 k x 4 m block -- k rows 4 kq + j -- so that its LDS writes are 4 consecutive k of one row;
            //  kq alternates every 8 lanes, see the k-loop's note on the 8-byte stores)
            const int krow = X3 ? 4 * (2 * wid + JF_X3_KQ(lane)) + (j & 3) : f / (BM / 4);
            const int c4 = X3 ? JF_X3_C4(lane) + 32 * (j >> 2) : f % (BM / 4);       // (BM = 256: a second block, 128 rows on)
            const int gm = m0 + c4 * 4;
            a_ptr[j] = P.A + gm;
            a_lim[j] = max(0, min(4, P.M - gm));   // valid elements along m
            a_k[j] = krow;
            a_lds[j] = krow * A_LD + c4 * 4;
        }
    }
#pragma unroll
    for (int j = 0; j < LB; ++j) {
        const int f = tid + j * NT;
        if (B_KC) {
            const int row = f / (BK / 4), c4 = f % (BK / 4);
            const int gn = n0 + row;
            const bool ok = gn < P.N;
            b_ptr[j] = P.B + (long long)(ok ? gn : 0) * P.ldb + c4 * 4;
            b_lim[j] = ok ? 1 : 0;
            b_k[j] = c4 * 4;
            b_lds[j] = row * B_LD + c4 * 4;
        } else {
            const int krow = X3 ? 4 * (2 * wid + JF_X3_KQ(lane)) + j : f / (BN / 4);
            const int c4 = X3 ? JF_X3_C4(lane) : f % (BN / 4);
            const int gn = n0 + c4 * 4;
            b_ptr[j] = P.B + gn;
            b_lim[j] = max(0, min(4, P.N - gn));
            b_k[j] = krow;
            b_lds[j] = krow * B_LD + c4 * 4;
        }
    }

    // ---- FAST path: buffer descriptors (wave-uniform) and per-thread byte offsets ----
    // A vector instruction beside the MFMAs costs the matrix pipe ~2.5 cycles whatever the occupancy (tools/micro/mfma_valu.hip:
    // 155 TF with none, 144 with 2 per MFMA, 133 with 4; scalar instructions and LDS reads are free), so the k-loop keeps its
    // per-thread offsets CONSTANT: the tile's position along K is the loads' SCALAR offset, the rows beyond M / N are an
    // out-of-range offset from the start, and a load costs one compare + select (k >= K -> out of range -> 0).  (The select, not
    // num_records, bounds k: whether the range check sees the scalar offset differs between descriptions of the hardware; the
    // out-of-range constant is out of range either way.)
    unsigned a_vo[LA], b_vo[LB];
    __amdgpu_buffer_rsrc_t a_rs, b_rs;
    if (FAST) {
        a_rs = __builtin_amdgcn_make_buffer_rsrc((void*)P.A, 0, (int)P.a_bytes, 0x00020000);
        b_rs = __builtin_amdgcn_make_buffer_rsrc((void*)P.B, 0, (int)P.b_bytes, 0x00020000);
#pragma unroll
        for (int j = 0; j < LA; ++j)
            a_vo[j] = a_lim[j] ? (unsigned)((a_ptr[j] - P.A) * 4) + (A_KC ? 0u : (unsigned)a_k[j] * (unsigned)P.lda * 4u) : JAMIE_OOB;
#pragma unroll
        for (int j = 0; j < LB; ++j)
            b_vo[j] = b_lim[j] ? (unsigned)((b_ptr[j] - P.B) * 4) + (B_KC ? 0u : (unsigned)b_k[j] * (unsigned)P.ldb * 4u) : JAMIE_OOB;
    }
    // (running scalar offsets, advanced by every load_tile call -- calls come in k order, one BK apart)
    const unsigned a_step = __builtin_amdgcn_readfirstlane((A_KC ? (unsigned)BK : (unsigned)BK * (unsigned)P.lda) * 4u);
    const unsigned b_step = __builtin_amdgcn_readfirstlane((B_KC ? (unsigned)BK : (unsigned)BK * (unsigned)P.ldb) * 4u);
    unsigned a_so = __builtin_amdgcn_readfirstlane((unsigned)(kbeg / BK) * a_step);
    unsigned b_so = __builtin_amdgcn_readfirstlane((unsigned)(kbeg / BK) * b_step);
    float4 ra[LA], rb[LB];
    auto load_tile = [&](int k0) {
        if (FAST) {
            const int kleft = kend - k0;         // (scalar; one compare + select per load: only the last k-tile's lanes beyond K)
#pragma unroll
            for (int j = 0; j < LA; ++j) ra[j] = buf_ld4(a_rs, a_k[j] < kleft ? a_vo[j] : JAMIE_OOB, a_so);
#pragma unroll
            for (int j = 0; j < LB; ++j) rb[j] = buf_ld4(b_rs, b_k[j] < kleft ? b_vo[j] : JAMIE_OOB, b_so);
            a_so += a_step; b_so += b_step;
            return;
        }
#pragma unroll
        for (int j = 0; j < LA; ++j) {
            if (A_KC) {
                const int k = k0 + a_k[j];
                const int nv = a_lim[j] ? max(0, min(4, kend - k)) : 0;
                ra[j] = ld4_guard(a_ptr[j] + k0, nv, avec);
            } else {
                const int k = k0 + a_k[j];
                const int nv = (k < kend) ? a_lim[j] : 0;
                ra[j] = ld4_guard(a_ptr[j] + (long long)k * P.lda, nv, avec);
            }
        }
#pragma unroll
        for (int j = 0; j < LB; ++j) {
            if (B_KC) {
                const int k = k0 + b_k[j];
                const int nv = b_lim[j] ? max(0, min(4, kend - k)) : 0;
                rb[j] = ld4_guard(b_ptr[j] + k0, nv, bvec);
            } else {
                const int k = k0 + b_k[j];
                const int nv = (k < kend) ? b_lim[j] : 0;
                rb[j] = ld4_guard(b_ptr[j] + (long long)k * P.ldb, nv, bvec);
            }
        }
    };
    auto store_tile = [&](int buf, int k0) {
        float* As = smem + buf * (A_SZ + B_SZ);
        float* Bs = As + A_SZ;
#pragma unroll
        for (int j = 0; j < LA; ++j) {
            float4 v = ra[j];
            if (FAST == 1) v = mask4(v, A_KC ? kend - (k0 + a_k[j]) : a_lim[j]);   // in-row tails (k or m)
            *reinterpret_cast<float4*>(&As[a_lds[j]]) = v;
        }
#pragma unroll
        for (int j = 0; j < LB; ++j) {
            float4 v = rb[j];
            if (FAST == 1) v = mask4(v, B_KC ? kend - (k0 + b_k[j]) : b_lim[j]);
            *reinterpret_cast<float4*>(&Bs[b_lds[j]]) = v;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    if constexpr (X3 != 0) {
        // ---- bf16x3 k-loop (four waves, one per SIMD; one workgroup per CU) ----
        // BM = 128: wave tiles of 64 x 64, three LDS stages; BM = 256: wave tiles of 128 x 64 (the cut per MFMA falls by a quarter,
        // a fragment read serves more MFMAs, a tile's prologue and stores are spread over twice the work), two LDS stages.
        // A plane is [rows][32 k] bf16 in 64-byte rows.  16-byte chunk c (8 k) of row R lives at chunk c ^ ((R >> 2) & 3) of the
        // 64-byte block R ^ ((R >> 4) & 3): (a) the fragment read -- lane (r, h) takes chunk 2 s + h of row R0 + r, 16 consecutive
        // rows per LDS cycle -- meets 16 different 16-byte slots of a 256-byte line; (b) a K-contiguous operand's writes (8 lanes per
        // row, 8 rows per instruction) fill two whole lines; (c) a row-contiguous operand's writes -- a thread holds 4 k x 4 m; an
        // 8-byte store is serviced 16 contiguous lanes at a time over 32 banks (MI355X_MICROARCH.md, LDS table): lanes 0-7 take rows
        // 4 c + e of 8 consecutive c, lanes 8-15 the other 8-byte half (the next 4 k) of the same chunks -- 16 different 8-byte
        // words of a 128-byte window (the block term varies with c >> 2, the chunk term with c & 3).  (With the halves 16 lanes apart
        // the counters showed a third of the dW launch's LDS cycles as bank conflicts: r05_pmc_step_f32_bf16x3.txt.)  No padding,
        // no transposed reads.
        typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
        typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
        constexpr int NST = BM == 128 ? 3 : 2;                     // LDS stages
        constexpr int LAG = NST - 1;                               // step kt cuts tile kt + LAG
        constexpr int PL_A = BM * 64, PL_B = BN * 64, ST_SZ = 3 * (PL_A + PL_B);        // bytes: one plane of A / of B, one stage
        constexpr int NLD = LA + LB;                               // float4 loads (= units of the cut) per thread and tile
        constexpr int MB = 6 * TM * TN;                            // MFMAs per k-block of 16
        constexpr int NF = 3 * (TM + TN);                          // fragments per k-block
        constexpr int NPH = 6 * NLD;                               // phases of a tile's cut: one per MFMA gap, from gap 0
        constexpr int BAR = NST == 3 ? MB + 3 : NPH + 1;           // the step's barrier sits behind MFMA number BAR
        static_assert(LB == 4 && (LA == 4 || LA == 8) && TN == 2 && NPH <= 2 * MB && BAR + NF + 4 <= 2 * MB && NF <= MB - NLD,
                      "bf16x3: gap budget of a k-step");
        unsigned char* const lds = reinterpret_cast<unsigned char*>(smem);
        auto img = [](int R, int c, int half) __attribute__((always_inline)) { return ((R ^ ((R >> 4) & 3)) << 6) + ((c ^ ((R >> 2) & 3)) << 4) + (half << 3); };
        const int rc4 = JF_X3_C4(lane), rkq = 2 * wid + JF_X3_KQ(lane);      // (the row-contiguous assignment above)
        int a_w[LA], b_w[LB];
#pragma unroll
        for (int u = 0; u < LA; ++u) {
            const int f = tid + u * NT;
            a_w[u] = A_KC ? img(f >> 3, (f & 7) >> 1, f & 1) : img(4 * (rc4 + 32 * (u >> 2)) + (u & 3), rkq >> 1, rkq & 1);
        }
#pragma unroll
        for (int u = 0; u < LB; ++u) {
            const int f = tid + u * NT;
            b_w[u] = B_KC ? img(f >> 3, (f & 7) >> 1, f & 1) : img(4 * rc4 + u, rkq >> 1, rkq & 1);
        }
        // one unit = 4 consecutive k of one tile row: cut into the three planes, three 8-byte LDS writes -- in SIX phases of at most 6
        // instructions, one behind each MFMA (state between the phases: cx / chi / cmid / clo).  An MFMA holds the SIMD's vector
        // issue for 8 of its 32 cycles and every vector instruction for 4: what exceeds 24 cycles in a gap is exposed
        // (MI355X_MICROARCH.md, cycle constants).
        float cx[4];
        u32x2 chi, cmid, clo;
        // (the subtractions as v_pk_add_f32 on pairs: 4 instructions fewer per unit, no faster -- 78.4 against 76.0 us on the d -> 2d shapes)
        auto cut_phase = [&](auto phc, unsigned char* base, int plane_sz, float x0, float x1, float x2, float x3) __attribute__((always_inline)) {
            constexpr int PH = decltype(phc)::value;
            constexpr unsigned HI = 0xFFFF0000u, SEL = 0x07060302u;
            if constexpr (PH == 0) {
                chi.x = __builtin_amdgcn_perm(__float_as_uint(x1), __float_as_uint(x0), SEL);
                chi.y = __builtin_amdgcn_perm(__float_as_uint(x3), __float_as_uint(x2), SEL);
                cx[0] = x0 - __uint_as_float(__float_as_uint(x0) & HI); cx[1] = x1 - __uint_as_float(__float_as_uint(x1) & HI);
            } else if constexpr (PH == 1) {
                cx[2] = x2 - __uint_as_float(__float_as_uint(x2) & HI); cx[3] = x3 - __uint_as_float(__float_as_uint(x3) & HI);
                cmid.x = __builtin_amdgcn_perm(__float_as_uint(cx[1]), __float_as_uint(cx[0]), SEL);
            } else if constexpr (PH == 2) {
                cmid.y = __builtin_amdgcn_perm(__float_as_uint(cx[3]), __float_as_uint(cx[2]), SEL);
                cx[0] = cx[0] - __uint_as_float(__float_as_uint(cx[0]) & HI); cx[1] = cx[1] - __uint_as_float(__float_as_uint(cx[1]) & HI);
            } else if constexpr (PH == 3) {
                cx[2] = cx[2] - __uint_as_float(__float_as_uint(cx[2]) & HI); cx[3] = cx[3] - __uint_as_float(__float_as_uint(cx[3]) & HI);
                clo.x = __builtin_amdgcn_perm(__float_as_uint(cx[1]), __float_as_uint(cx[0]), SEL);
            } else if constexpr (PH == 4) {
                clo.y = __builtin_amdgcn_perm(__float_as_uint(cx[3]), __float_as_uint(cx[2]), SEL);
                *reinterpret_cast<u32x2*>(base) = chi;
            } else {
                *reinterpret_cast<u32x2*>(base + plane_sz) = cmid;
                *reinterpret_cast<u32x2*>(base + 2 * plane_sz) = clo;
            }
        };
        auto cut_store = [&](unsigned char* base, int plane_sz, auto phc, float x0, float x1, float x2, float x3) __attribute__((always_inline)) {
            constexpr int PH = decltype(phc)::value;          // 0 .. 5: that phase; 6: all of them
            if constexpr (PH == 6)
                jf_static_for<0, 6>([&](auto pc) __attribute__((always_inline)) { cut_phase(pc, base, plane_sz, x0, x1, x2, x3); });
            else cut_phase(phc, base, plane_sz, x0, x1, x2, x3);
        };
        auto comp = [](const float4& v, int e) __attribute__((always_inline)) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; };
        // Staging registers.  BM = 128: two sets, tile t rides in set t & 1 from its request at the top of a step, a whole k-step
        // before its cut starts (with one set and half a step the loop ran at the memory latency, 1.9 us per k-step).  BM = 256: ONE set
        // (two would be 96 registers beside 128 accumulators and up to 96 of fragments: the buffer descriptors ended up in vector
        // registers, a readfirstlane loop in front of every load): a float4 is requested again -- for the tile after -- in the gap
        // behind the last phase that reads it, so it still has a whole step to arrive.
        constexpr int NSET = BM == 128 ? 2 : 1;
        float4 xa[NSET][LA], xb[NSET][LB];
        auto load_one = [&](auto sc, auto jc, int k0) __attribute__((always_inline)) {          // load j of a tile (0 .. LA-1: A, then B); the last of an operand advances its k
            constexpr int S = decltype(sc)::value, J = decltype(jc)::value;
            const int kleft = kend - k0;
            if constexpr (J < LA) xa[S][J] = buf_ld4(a_rs, a_k[J] < kleft ? a_vo[J] : JAMIE_OOB, a_so);
            else xb[S][J - LA] = buf_ld4(b_rs, b_k[J - LA] < kleft ? b_vo[J - LA] : JAMIE_OOB, b_so);
            if constexpr (J == LA - 1) a_so += a_step;
            if constexpr (J == NLD - 1) b_so += b_step;
        };
        // the gap in which load j of the NEXT tile to be cut goes out: two sets: gap j; one set: behind phase 1 of the unit (K-contiguous
        // operand) or of the block's last unit (row-contiguous: its four float4 are read by all four units of the block) that reads it
        auto load_gap = [](int j) __attribute__((always_inline)) {
            if (NSET == 2) return j;
            const bool kc = j < LA ? A_KC : B_KC;
            return kc ? 6 * j + 2 : 6 * (j | 3) + 2 + (j & 3);       // (j counts A's then B's float4: unit numbers)
        };
        auto load_x3 = [&](auto sc, int k0) __attribute__((always_inline)) { jf_static_for<0, NLD>([&](auto jc) __attribute__((always_inline)) { load_one(sc, jc, k0); }); };
        // unit u of the staged tile (0 .. LA-1: A, then B) into stage `st`; k0 = the tile's first k (in-row tails of the FAST == 1 instance)
        auto unit = [&](unsigned char* st, int k0, auto sc, auto uc, auto phc) __attribute__((always_inline)) {
            constexpr int U = decltype(uc)::value, S = decltype(sc)::value;
            if constexpr (U < LA) {
                constexpr int e = U & 3, q = U & ~3;             // (row-contiguous: component e of the four k rows of block q / 4)
                if constexpr (A_KC) {
                    float4 v = xa[S][U];
                    if (FAST == 1) v = mask4(v, kend - (k0 + a_k[U]));
                    cut_store(st + a_w[U], PL_A, phc, v.x, v.y, v.z, v.w);
                } else {
                    const bool ok = FAST != 1 || e < a_lim[q];
                    cut_store(st + a_w[U], PL_A, phc, ok ? comp(xa[S][q], e) : 0.f, ok ? comp(xa[S][q + 1], e) : 0.f,
                              ok ? comp(xa[S][q + 2], e) : 0.f, ok ? comp(xa[S][q + 3], e) : 0.f);
                }
            } else {
                constexpr int V = U - LA, e = V & 3;
                if constexpr (B_KC) {
                    float4 v = xb[S][V];
                    if (FAST == 1) v = mask4(v, kend - (k0 + b_k[V]));
                    cut_store(st + 3 * PL_A + b_w[V], PL_B, phc, v.x, v.y, v.z, v.w);
                } else {
                    const bool ok = FAST != 1 || e < b_lim[0];
                    cut_store(st + 3 * PL_A + b_w[V], PL_B, phc, ok ? comp(xb[S][0], e) : 0.f, ok ? comp(xb[S][1], e) : 0.f,
                              ok ? comp(xb[S][2], e) : 0.f, ok ? comp(xb[S][3], e) : 0.f);
                }
            }
        };
        typedef std::integral_constant<int, 6> PHALL;
        auto cut_tile = [&](unsigned char* st, int k0, auto sc) __attribute__((always_inline)) { jf_static_for<0, NLD>([&](auto uc) __attribute__((always_inline)) { unit(st, k0, sc, uc, PHALL{}); }); };
        int a_r[TM], a_s[TM], b_r[TN], b_s[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) { const int R = wm0 + i * 32 + r; a_r[i] = (R ^ ((R >> 4) & 3)) << 6; a_s[i] = (R >> 2) & 3; }
#pragma unroll
        for (int j = 0; j < TN; ++j) { const int R = wn0 + j * 32 + r; b_r[j] = (R ^ ((R >> 4) & 3)) << 6; b_s[j] = (R >> 2) & 3; }
        bf16x8 fa[2][TM][3], fb[2][TN][3];
        auto read_one = [&](auto setc, auto fc, const unsigned char* st, int s16) __attribute__((always_inline)) {      // fragment f of the NF of a k-block
            constexpr int SET = decltype(setc)::value, F = decltype(fc)::value;
            if constexpr (F < 3 * TM) {
                constexpr int i = F / 3, pl = F % 3;
                fa[SET][i][pl] = *reinterpret_cast<const bf16x8*>(st + pl * PL_A + a_r[i] + (((2 * s16 + h) ^ a_s[i]) << 4));
            } else {
                constexpr int j = (F - 3 * TM) / 3, pl = (F - 3 * TM) % 3;
                fb[SET][j][pl] = *reinterpret_cast<const bf16x8*>(st + 3 * PL_A + pl * PL_B + b_r[j] + (((2 * s16 + h) ^ b_s[j]) << 4));
            }
        };
        typedef std::integral_constant<int, 0> S0;
        typedef std::integral_constant<int, 1> S1;
        // One k-block: MFMA m (the six products of the block's TM x TN tiles, smallest first), then its gap -- one wave per SIMD, so
        // what the matrix pipe does not get from THIS wave's instruction stream it does not get, and every piece of the step's
        // other work is pinned into a gap.  With g = the MFMA's number in the step: phase g of the cut of tile kt+LAG (g < NPH);
        // load g of tile kt+LAG+1 (g < NLD); fragment g - (MB - NF) of this tile's second block (MB - NF <= g < MB); behind MFMA BAR
        // the step's barrier (the MFMAs issued before it are queued while the wave waits for its LDS writes and for the others);
        // fragment g - BAR - 1 of the NEXT tile's first block (BAR < g <= BAR + NF).  (Measured on the way, 128 x 128, per k-step of
        // 48 MFMAs = 1536 pipe cycles: three MFMAs then a whole unit, all units in the first k-block, two LDS stages: 1.9 us; a third
        // of a unit behind every MFMA of the first k-block: 1.45; three stages, a third behind every second MFMA: 1.3; a sixth behind
        // every MFMA: 1.05; loads and fragment reads in the gaps too: 1.0.  sched_group_barrier pipelines held in one of the two
        // unrolled steps only.)
        auto block = [&](auto setc, auto pc, auto cutc, auto morec, unsigned char* cur, unsigned char* nx1, unsigned char* nxc, int kt) __attribute__((always_inline)) {
            constexpr int SET = decltype(setc)::value, PAR = decltype(pc)::value;
            constexpr bool CUT = decltype(cutc)::value, MORE = decltype(morec)::value;
            constexpr int CS = NSET == 2 ? PAR ^ (LAG & 1) : 0;   // staging set of tile kt + LAG
            constexpr int LS = NSET == 2 ? CS ^ 1 : 0;           // ... the set tile kt + LAG + 1 is requested into
            jf_static_for<0, MB>([&](auto mc) __attribute__((always_inline)) {
                constexpr int m = decltype(mc)::value, g = SET * MB + m, q = m / (TM * TN), i = (m % (TM * TN)) / TN, j = m % TN;
                constexpr int QA[6] = {2, 0, 1, 1, 0, 0}, QB[6] = {0, 2, 1, 0, 1, 0};
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[SET][i][QA[q]], fb[SET][j][QB[q]], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (CUT) {
                    if constexpr (g < NPH)
                        unit(nxc, kbeg + (kt + LAG) * BK, std::integral_constant<int, CS>{}, std::integral_constant<int, g / 6>{}, std::integral_constant<int, g % 6>{});
                    jf_static_for<0, NLD>([&](auto jc) __attribute__((always_inline)) {
                        if constexpr (load_gap(decltype(jc)::value) == g) load_one(std::integral_constant<int, LS>{}, jc, kbeg + (kt + LAG + 1) * BK);
                    });
                }
                if constexpr (SET == 0 && m >= MB - NF) read_one(S1{}, std::integral_constant<int, m - (MB - NF)>{}, cur, 1);
                if constexpr (MORE) {
                    if constexpr (g == BAR) {
                        __builtin_amdgcn_s_waitcnt(0xC07F);       // lgkmcnt(0): my LDS writes so far are out, my second block's fragments in
                        __builtin_amdgcn_s_barrier();
                    }
                    if constexpr (g > BAR && g <= BAR + NF) read_one(S0{}, std::integral_constant<int, g - BAR - 1>{}, nx1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        // Tile t lives in LDS stage t % NST and travels in register set t & 1: requested a whole step before its cut, cut during step
        // t - LAG (one phase per gap), multiplied in step t.  ONE barrier per step: a wave behind barrier kt has finished every write
        // of the tile(s) cut up to there and all its reads of tile kt-1 lie in front of barrier kt-1.  Three stages: the cut of tile
        // kt+2 fills the WHOLE step (both k-blocks) and the barrier sits early in the second block; two stages (BM = 256, 144 KB): the
        // cut of tile kt+1 ends before the barrier, which sits behind it, three quarters into the step.
        if (nk > 0) {                                             // (beyond the K slice: out-of-range offsets -- zeros, no traffic)
            load_x3(S0{}, kbeg);
            if constexpr (LAG == 2) {
                load_x3(S1{}, kbeg + BK);
                cut_tile(lds, kbeg, S0{});
                load_x3(S0{}, kbeg + 2 * BK);
                cut_tile(lds + ST_SZ, kbeg + BK, S1{});
            } else {
                cut_tile(lds, kbeg, S0{});
                load_x3(S0{}, kbeg + BK);
            }
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);           // lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        JF_STAMP(1);
        if (nk > 0) jf_static_for<0, NF>([&](auto fc) __attribute__((always_inline)) { read_one(S0{}, fc, lds, 0); });
        int s_cur = 0;                                // stage of tile kt
        // CUT: tile kt+LAG exists; MORE: tile kt+1 exists
        auto step = [&](int kt, auto pc, auto cutc, auto morec) __attribute__((always_inline)) {
            const int s_n1 = s_cur == NST - 1 ? 0 : s_cur + 1, s_n2 = s_n1 == NST - 1 ? 0 : s_n1 + 1;
            unsigned char* const cur = lds + s_cur * ST_SZ;
            unsigned char* const nx1 = lds + s_n1 * ST_SZ;
            unsigned char* const nxc = lds + (LAG == 2 ? s_n2 : s_n1) * ST_SZ;
            __builtin_amdgcn_sched_barrier(0);
            block(S0{}, pc, cutc, morec, cur, nx1, nxc, kt);
            block(S1{}, pc, cutc, morec, cur, nx1, nxc, kt);
            s_cur = s_n1;
        };
        typedef std::true_type T_;
        typedef std::false_type F_;
        int kt = 0;
        for (; kt + LAG + 1 < nk; kt += 2) {          // (straight-line pairs: a branch inside made hipcc keep the accumulators in two register sets)
            step(kt, S0{}, T_{}, T_{});
            step(kt + 1, S1{}, T_{}, T_{});
        }
        const int rest = nk - kt;
        if constexpr (LAG == 2) {
            if (rest == 3) {
                step(kt, S0{}, T_{}, T_{});
                step(kt + 1, S1{}, F_{}, T_{});
                step(kt + 2, S0{}, F_{}, F_{});
            } else if (rest == 2) {
                step(kt, S0{}, F_{}, T_{});
                step(kt + 1, S1{}, F_{}, F_{});
            } else if (rest == 1) {
                step(kt, S0{}, F_{}, F_{});
            }
        } else {
            if (rest == 2) {
                step(kt, S0{}, T_{}, T_{});
                step(kt + 1, S1{}, F_{}, F_{});
            } else if (rest == 1) {
                step(kt, S0{}, F_{}, F_{});
            }
        }
    } else {
    if (nk > 0) {
        load_tile(kbeg);
        store_tile(0, kbeg);
    }
    if constexpr (MID) {
        // One barrier per k-step, between the MFMAs of the last-but-one and the last k-group.  A k-step of the loop below leaves
        // the matrix pipe idle twice (in-kernel stamps, tools/stamp_gemm_f32.py: ~250 ns after its barrier until the first
        // fragments of the new tile are back from LDS, ~240 ns in front of it while the next tile is written: 20 % of a k-step
        // of one workgroup, and two workgroups per CU only win back half).  Here the next tile is written to LDS right behind
        // the FIRST k-group's MFMAs (its global loads were issued a k-step earlier; the loads of the tile after it go out as
        // soon as the staging registers are free), every fragment read of the current tile is issued before the barrier
        // (k-group g+1 is fetched under the MFMAs of g), and the first fragments of the NEXT tile are fetched right behind it,
        // under the last k-group's MFMAs.  Hazards: the buffer written in k-step kt was last read in k-step kt-1, whose reads
        // all completed in front of that step's barrier; the buffer read behind the barrier was written in front of it.
        constexpr int G = BK / 8;
        static_assert(G % 2 == 0, "k-groups alternate between two fragment sets");
        if (nk > 1) load_tile(kbeg + BK);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        JF_STAMP(1);
        float af[2][TM][4], bf[2][TN][4];
        auto read_frags = [&](int buf, const float* As, const float* Bs, int kk) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if (A_KC) {
                    const float4 v = *reinterpret_cast<const float4*>(&As[(wm0 + i * 32 + r) * A_LD + kk + 4 * h]);
                    af[buf][i][0] = v.x; af[buf][i][1] = v.y; af[buf][i][2] = v.z; af[buf][i][3] = v.w;
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s) af[buf][i][s] = As[(kk + 4 * h + s) * A_LD + wm0 + i * 32 + r];
                }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (B_KC) {
                    const float4 v = *reinterpret_cast<const float4*>(&Bs[(wn0 + j * 32 + r) * B_LD + kk + 4 * h]);
                    bf[buf][j][0] = v.x; bf[buf][j][1] = v.y; bf[buf][j][2] = v.z; bf[buf][j][3] = v.w;
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s) bf[buf][j][s] = Bs[(kk + 4 * h + s) * B_LD + wn0 + j * 32 + r];
                }
            }
        };
        if (nk > 0) read_frags(0, smem, smem + A_SZ, 0);
        for (int kt = 0; kt < nk; ++kt) {
            const int cur = kt & 1;
            const bool more = kt + 1 < nk;
            const float* As = smem + cur * (A_SZ + B_SZ);
            const float* Bs = As + A_SZ;
            const float* An = smem + (cur ^ 1) * (A_SZ + B_SZ);
            JF_KSTAMP(kt, nk, 0);
#pragma unroll
            for (int gk = 0; gk < G; ++gk) {
                __builtin_amdgcn_sched_barrier(0);
                if (gk + 1 < G) read_frags((gk + 1) & 1, As, Bs, (gk + 1) * 8);
                else if (more) read_frags(0, An, An + A_SZ, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[gk & 1][i][s], bf[gk & 1][j][s], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (gk == 0) {
                    if (more) store_tile(cur ^ 1, kbeg + (kt + 1) * BK);
                    __builtin_amdgcn_sched_barrier(0);
                    if (kt + 2 < nk) load_tile(kbeg + (kt + 2) * BK);
                    JF_KSTAMP(kt, nk, 1);
                }
                if (gk == G - 2) {
                    JF_KSTAMP(kt, nk, 2);
                    // (the builtin, not inline asm: the compiler's wait-count pass then knows that the last k-group's fragments
                    //  have arrived and waits with lgkmcnt(2), not (0), in front of its MFMAs -- behind the next tile's first reads)
                    __builtin_amdgcn_s_waitcnt(0xC07F);           // lgkmcnt(0)
                    __builtin_amdgcn_s_barrier();
                    JF_KSTAMP(kt, nk, 3);
                }
            }
        }
    } else {
    __syncthreads();
    JF_STAMP(1);
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nk;
        JF_KSTAMP(kt, nk, 0);
        if (more && !(JAMIE_GEMM_ABL & 1)) load_tile(kbeg + (kt + 1) * BK);
        const float* As = smem + cur * (A_SZ + B_SZ);
        const float* Bs = As + A_SZ;
        // fragments are fetched one k-group (8 k) ahead of the MFMAs that consume them, so the LDS latency
        // of group g+1 runs under the 4*TM*TN MFMAs of group g
        float af[2][TM][4], bf[2][TN][4];
        auto read_frags = [&](int buf, int kk) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if (A_KC) {
                    const float4 v = *reinterpret_cast<const float4*>(&As[(wm0 + i * 32 + r) * A_LD + kk + 4 * h]);
                    af[buf][i][0] = v.x; af[buf][i][1] = v.y; af[buf][i][2] = v.z; af[buf][i][3] = v.w;
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s) af[buf][i][s] = As[(kk + 4 * h + s) * A_LD + wm0 + i * 32 + r];
                }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (B_KC) {
                    const float4 v = *reinterpret_cast<const float4*>(&Bs[(wn0 + j * 32 + r) * B_LD + kk + 4 * h]);
                    bf[buf][j][0] = v.x; bf[buf][j][1] = v.y; bf[buf][j][2] = v.z; bf[buf][j][3] = v.w;
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s) bf[buf][j][s] = Bs[(kk + 4 * h + s) * B_LD + wn0 + j * 32 + r];
                }
            }
        };
#if JAMIE_GEMM_ABL & 16
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int i = 0; i < TM; ++i) asm volatile("v_mov_b32 %0, 1.0" : "=v"(af[q][i][s]));
#pragma unroll
                for (int j = 0; j < TN; ++j) asm volatile("v_mov_b32 %0, 1.0" : "=v"(bf[q][j][s]));
            }
#else
        read_frags(0, 0);
#endif
#pragma unroll
        for (int gk = 0; gk < BK / 8; ++gk) {
            __builtin_amdgcn_sched_barrier(0);   // pin: reads of group g+1 are ISSUED before the MFMAs of group g
            if (gk + 1 < BK / 8 && !(JAMIE_GEMM_ABL & 16)) read_frags((gk + 1) & 1, (gk + 1) * 8);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                    {
#if JAMIE_GEMM_ABL & 2
                        asm volatile("" ::"v"(af[gk & 1][i][s]), "v"(bf[gk & 1][j][s]));
#else
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[gk & 1][i][s], bf[gk & 1][j][s], acc[i][j], 0, 0, 0);
#endif
                    }
        }
        __builtin_amdgcn_sched_barrier(0);      // keep the masks / LDS writes (and their vmcnt wait) after the MFMAs
        JF_KSTAMP(kt, nk, 1);
        if (more && !(JAMIE_GEMM_ABL & 4)) store_tile(cur ^ 1, kbeg + (kt + 1) * BK);
        JF_KSTAMP(kt, nk, 2);
        if (!(JAMIE_GEMM_ABL & 8)) __syncthreads();
        JF_KSTAMP(kt, nk, 3);
    }
    }
    }

    JF_STAMP(2);
    // ---- epilogue.  C/D map of the 32x32 tile: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5) ----
    float* Cout = P.C + (long long)ks * P.slab_stride;
    const bool add_bias = (P.bias != nullptr) && ks == 0;
    float local = 0.f;
    // Plain stores (split-K slabs, weight gradients): one lean path.  Beside a co-resident workgroup that saturates the matrix
    // pipe a vector or scalar instruction of this epilogue gets an issue slot about once per MFMA, and the general loop below
    // spends ~8 vector instructions and ~10 branches per element (stamps: 12-18 us of epilogue beside a computing workgroup, 4 us
    // alone) while the tile's slot on the CU stays taken.  Here an element is one add (bias), one FMA (sum of squares) and one
    // buffer store whose row offset is scalar; rows beyond M only cost a select in the tiles that have any.
    if (FAST && (P.epi == JAMIE_EPI_STORE || P.epi == JAMIE_EPI_BN_EVAL) && !P.accumulate && P.c_bytes != 0) {
        const __amdgpu_buffer_rsrc_t c_rs = __builtin_amdgcn_make_buffer_rsrc((void*)Cout, 0, (int)P.c_bytes, 0x00020000);
        const bool edge = m0 + BM > P.M || n0 + BN > P.N;
        const unsigned ldc4 = (unsigned)P.ldc * 4u;
        auto emit = [&](auto nt_c, auto edge_c, auto bn_c) {            // (straight-line instances behind scalar branches)
            constexpr bool NT_ST = decltype(nt_c)::value, EDGE = decltype(edge_c)::value, EVAL_BN = decltype(bn_c)::value;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn0 + j * 32 + r;
                const bool nok = n < P.N;
                const float bv = (add_bias && nok) ? P.bias[n] : 0.f;
                float e_mean = 0.f, e_scale = 1.f, e_shift = 0.f;
                if (EVAL_BN && nok) {       // eval BatchNorm + LeakyReLU, the general loop's operation order
                    e_mean = P.aux0[n];
                    e_scale = rsqrtf(P.aux1[n] + P.eps) * P.aux2[n];
                    e_shift = P.aux3[n];
                }
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int mrow = m0 + wm0 + i * 32 + 4 * h;                       // + (e & 3) + 8 * (e >> 2)
                    const unsigned voff = nok ? (unsigned)mrow * ldc4 + (unsigned)n * 4u : JAMIE_OOB;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int rel = (e & 3) + 8 * (e >> 2);
                        float v = acc[i][j][e] + bv;
                        if (EVAL_BN) {
                            v = (v - e_mean) * e_scale + e_shift;
                            v = v > 0.f ? v : P.slope * v;
                        }
                        unsigned vo = voff;
                        if (EDGE) vo = (mrow + rel < P.M) ? voff : JAMIE_OOB;
#ifdef JAMIE_STORE_SOFF       // (A/B: round 4's form, the row offset as the store's SCALAR offset)
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), c_rs, (int)vo, (int)((unsigned)rel * ldc4), NT_ST ? 2 : 0);
#else
                        // (row offset in the VECTOR offset: no buffer store with an SGPR offset anywhere in the product, see gemm_bf16.hip)
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), c_rs, (int)((EDGE && vo == JAMIE_OOB) ? vo : vo + (unsigned)rel * ldc4), 0, NT_ST ? 2 : 0);
#endif
                        if (!EVAL_BN) {
                            if (EDGE) local += (vo != JAMIE_OOB) ? v * v : 0.f;
                            else local += v * v;                  // (interior tiles: every element counts)
                        }
                    }
                }
            }
        };
        if (P.epi == JAMIE_EPI_BN_EVAL) {
            asm volatile("; eval BatchNorm epilogue");
            if (edge) { asm volatile("; edge tile"); emit(std::false_type{}, std::true_type{}, std::true_type{}); }
            else emit(std::false_type{}, std::false_type{}, std::true_type{});
        } else if (P.store_nt) {
            asm volatile("; nt stores");
            if (edge) { asm volatile("; edge tile"); emit(std::true_type{}, std::true_type{}, std::false_type{}); }
            else emit(std::true_type{}, std::false_type{}, std::false_type{});
        } else {
            asm volatile("; plain stores");
            if (edge) { asm volatile("; edge tile"); emit(std::false_type{}, std::true_type{}, std::false_type{}); }
            else emit(std::false_type{}, std::false_type{}, std::false_type{});
        }
        if (P.partial != nullptr) {
            const float tot = block_sum(local, red);
            if (tid == 0) P.partial[t] = tot;
        }
#ifdef JAMIE_GEMMB_STAMP
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        JF_STAMP(3);
        JF_STAMPV(7, __builtin_amdgcn_s_memtime());
#endif
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn0 + j * 32 + r;
        if (n >= P.N) continue;
        const float bv = add_bias ? P.bias[n] : 0.f;
        float e_mean = 0.f, e_scale = 1.f, e_shift = 0.f;
        if (P.epi == JAMIE_EPI_BN_EVAL) {
            e_mean = P.aux0[n];
            e_scale = rsqrtf(P.aux1[n] + P.eps) * P.aux2[n];
            e_shift = P.aux3[n];
        }
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
                    if (P.store_nt) __builtin_nontemporal_store(v, cp);     // weight gradients: see gemm_bf16.hip
                    else *cp = v;
                    local += v * v;                                         // (used when `partial` is given: the clip norm)
                } else if (P.epi == JAMIE_EPI_MSE) {
                    const float d = v - P.aux0[(long long)m * P.aux_ld + n];
                    local += d * d;
                    *cp = d * P.scale;
                } else {  // JAMIE_EPI_BN_EVAL
                    const float y = (v - e_mean) * e_scale + e_shift;
                    *cp = y > 0.f ? y : P.slope * y;
                }
            }
        }
    }
    if (P.epi == JAMIE_EPI_MSE && P.partial != nullptr) {
        const float tot = block_sum(local, red);
        if (tid == 0) P.partial[t] = tot * P.pscale;
    }
    // weight gradients: the tile's sum of squares (fixed order) for the global-norm clip, as in the bf16 kernel -- the
    // separate pass over the 161 MB gradient buffer (27 us) is then only a pass over the ranges no GEMM writes
    if (P.epi == JAMIE_EPI_STORE && P.partial != nullptr) {
        const float tot = block_sum(local, red);
        if (tid == 0) P.partial[t] = tot;
    }
#ifdef JAMIE_GEMMB_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    JF_STAMP(3);
    JF_STAMPV(7, __builtin_amdgcn_s_memtime());

#endif
}


// ------------------------------------------------------------------------------------------------
// LDS-DMA variant for the NT layout (both operands K-contiguous: the forward Linear products and the eval path).
// The counters on the register-staged kernel above (tools/pmc_gemm_f32.sh) show the matrix pipe busy 61 % of the launch
// at full clock, and its ablations charge 20 % of the time to the VGPR -> LDS staging writes (16 ds_write_b128 per
// workgroup and k-step) and 6 % to exposed global loads.  Here whole k-tiles of 32 floats (128-byte rows) go
// global -> LDS directly (`global_load_lds_dwordx4`, 1 KiB = 8 rows per wave-instruction: no staging registers, no
// ds_write), NB buffers deep, with hand-counted `s_waitcnt vmcnt(N)` and a raw `s_barrier` (a __syncthreads() would
// drain vmcnt(0)).  The LDS image is lane-linear, so rows cannot be padded: 16-byte chunk c of row r sits at chunk
// c ^ ((r >> 1) & 7) -- applied to the SOURCE address of the DMA and to the fragment read -- which keeps the
// ds_read_b128 of the 32 rows of an MFMA operand conflict-free (the image of the bf16 kernel, gemm_bf16.hip).
// Rows beyond M / N re-read the last valid row (never stored); a partial last k-tile goes through registers, masked.
// ------------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int TAG, int NB>
__global__ __launch_bounds__(WM * WN * 64) void gemm_f32_dma_kernel(GemmGroup g) {
    constexpr int BK = 32, NW = WM * WN, NT = NW * 64;
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int A_SZ = BM * 128, B_SZ = BN * 128;      // bytes, unpadded 128-byte rows
    constexpr int PA = BM / 8 / NW, PB = BN / 8 / NW;    // 1-KiB pieces per wave
    constexpr int LA = BM * 8 / NT, LB = BN * 8 / NT;    // 16-byte chunks per thread (register tail path)
    constexpr int GL = PA + PB;
    static_assert(PA >= 1 && PB >= 1 && (BM / 8) % NW == 0 && (BN / 8) % NW == 0, "tile/wave mismatch");
    static_assert(GL * (NB - 1) <= 60 && NB >= 2 && NB <= 4, "vmcnt immediate");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NB * (A_SZ + B_SZ)];
    float* red = reinterpret_cast<float*>(smem);

    JF_STAMP(0);
    const int bid = blockIdx.x;
    const int xcd = bid & 7;
    int slot = bid >> 3;
    int pi = 0, t = 0, rot = 0;
    // (branch-free, every problem's tile count loaded up front -- unused problems hold 0: written as a loop of guarded
    //  iterations this was one dependent scalar-memory round trip and three branches per problem, 2-3 us of a tile's time on its CU
    //  slot before the first load, more beside a workgroup that saturates the matrix pipe)
    int ntl[JAMIE_MAX_GEMM_GROUP_F32];
#pragma unroll
    for (int i = 0; i < JAMIE_MAX_GEMM_GROUP_F32; ++i) ntl[i] = g.ntiles[i];
    bool found = false;
    const int n_prob = g.count;
#pragma unroll
    for (int i = 0; i < JAMIE_MAX_GEMM_GROUP_F32; ++i) {
        if (i >= n_prob) break;                    // (one scalar branch: the forward / dX launches hold 2 of 12 problems)
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
    const GemmDev& P = g.p[pi];          // (by value, as gemm_bf16.hip does: +5 us per fp32 step)
    // t -> (M tile, N tile, K slice) by multiply-high with reciprocals from the host (exact below 65536 tiles; 0: divide -- a
    // scalar integer division is ~30 instructions through the vector unit's reciprocal and back)
    int tm_i, tn_i, ks;
    if (P.inv_tm != 0) {
        const unsigned q1 = P.tiles_m == 1 ? (unsigned)t : __umulhi((unsigned)t, P.inv_tm);      // (2^32 / 1 + 1 does not fit)
        const unsigned q2 = P.tiles_n == 1 ? q1 : __umulhi(q1, P.inv_tn);
        tm_i = t - (int)q1 * P.tiles_m;
        tn_i = (int)q1 - (int)q2 * P.tiles_n;
        ks = (int)q2;
    } else {
        tm_i = t % P.tiles_m;
        tn_i = (t / P.tiles_m) % P.tiles_n;
        ks = t / (P.tiles_m * P.tiles_n);
    }
    const int m0 = tm_i * BM, n0 = tn_i * BN;
    const int kbeg = ks * P.kchunk;
    const int kend = min(P.K, kbeg + P.kchunk);
    const int nk = (kend - kbeg + BK - 1) / BK;
    const int nfull = (kend - kbeg) / BK;
    JF_STAMPV(4, pi * 1000 + nk);
    JF_STAMPV(5, __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) * 1000 + __builtin_amdgcn_s_getreg(((8 - 1) << 11) | (8 << 6) | 4));

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm0 = (wid / WN) * (TM * 32), wn0 = (wid % WN) * (TN * 32);
    const int r = lane & 31, h = lane >> 5;
    const int lrow = lane >> 3, pch = lane & 7;
    const float* a_src[PA]; const float* b_src[PB];
#pragma unroll
    for (int i = 0; i < PA; ++i) {
        const int row = 8 * (wid + NW * i) + lrow;
        a_src[i] = P.A + (long long)min(m0 + row, P.M - 1) * P.lda + ((pch ^ ((row >> 1) & 7)) * 4);
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
        const int row = 8 * (wid + NW * i) + lrow;
        b_src[i] = P.B + (long long)min(n0 + row, P.N - 1) * P.ldb + ((pch ^ ((row >> 1) & 7)) * 4);
    }
    typedef const void __attribute__((address_space(1)))* gptr_t;
    typedef void __attribute__((address_space(3)))* lptr_t;
    auto tail_ld = [&](const float* p, int k) {      // 4 floats of a row from column k, zero beyond kend
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k + 3 < kend) v = *reinterpret_cast<const float4*>(p + k);
        else {
            if (k < kend) v.x = p[k];
            if (k + 1 < kend) v.y = p[k + 1];
            if (k + 2 < kend) v.z = p[k + 2];
        }
        return v;
    };
    auto stage = [&](int buf, int kt) {
        unsigned char* As = smem + buf * (A_SZ + B_SZ);
        unsigned char* Bs = As + A_SZ;
        const int k0 = kbeg + kt * BK;
        if (kt < nfull) {
#pragma unroll
            for (int i = 0; i < PA; ++i)
                __builtin_amdgcn_global_load_lds((gptr_t)(a_src[i] + k0), (lptr_t)(As + (wid + NW * i) * 1024), 16, 0, 0);
#pragma unroll
            for (int i = 0; i < PB; ++i)
                __builtin_amdgcn_global_load_lds((gptr_t)(b_src[i] + k0), (lptr_t)(Bs + (wid + NW * i) * 1024), 16, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < LA; ++j) {
                const int f = tid + j * NT, row = f >> 3, c = f & 7;
                const float4 v = tail_ld(P.A + (long long)min(m0 + row, P.M - 1) * P.lda, k0 + c * 4);
                *reinterpret_cast<float4*>(As + row * 128 + ((c ^ ((row >> 1) & 7)) << 4)) = v;
            }
#pragma unroll
            for (int j = 0; j < LB; ++j) {
                const int f = tid + j * NT, row = f >> 3, c = f & 7;
                const float4 v = tail_ld(P.B + (long long)min(n0 + row, P.N - 1) * P.ldb, k0 + c * 4);
                *reinterpret_cast<float4*>(Bs + row * 128 + ((c ^ ((row >> 1) & 7)) << 4)) = v;
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
    // prologue: tiles 0 .. NB-2 in flight
#pragma unroll
    for (int u = 0; u < NB - 1; ++u)
        if (u < nk) stage(u, u);
    JF_STAMP(1);
    for (int kt = 0; kt < nk; ++kt) {
        // my pieces of tile kt have landed: DMA tiles younger than kt already issued = kt+1 .. kt+NB-2 (full tiles only)
        const int younger = max(0, min(kt + NB - 2, nfull - 1) - kt);
        if (kt >= nfull || younger == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GL) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NB > 3 ? 2 * GL : 0) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                 // tile kt visible to all; buffer (kt-1) % NB free
        __builtin_amdgcn_sched_barrier(0);
        if (kt + NB - 1 < nk) stage((kt + NB - 1) % NB, kt + NB - 1);
        const unsigned char* As = smem + (kt % NB) * (A_SZ + B_SZ);
        const unsigned char* Bs = As + A_SZ;
        float4 af[2][TM], bf[2][TN];
        auto read_frags = [&](int buf, int gk) {
            const int off = ((2 * gk + h) ^ swz) << 4;
#pragma unroll
            for (int i = 0; i < TM; ++i) af[buf][i] = *reinterpret_cast<const float4*>(As + (wm0 + i * 32 + r) * 128 + off);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[buf][j] = *reinterpret_cast<const float4*>(Bs + (wn0 + j * 32 + r) * 128 + off);
        };
        read_frags(0, 0);
#pragma unroll
        for (int gk = 0; gk < BK / 8; ++gk) {
            const float* a0 = reinterpret_cast<const float*>(&af[gk & 1][0]);
            const float* b0 = reinterpret_cast<const float*>(&bf[gk & 1][0]);
            // the prefetch goes AFTER the first MFMA of the group (hipcc puts an uncounted lgkmcnt(0) before an MFMA that
            // follows a pending LDS-DMA: the wait then only covers reads that have had a whole group to land)
            __builtin_amdgcn_sched_barrier(0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[0], b0[0], acc[0][0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (gk + 1 < BK / 8) read_frags((gk + 1) & 1, gk + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        if (s4 + i + j > 0)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(reinterpret_cast<const float*>(&af[gk & 1][i])[s4],
                                                                             reinterpret_cast<const float*>(&bf[gk & 1][j])[s4],
                                                                             acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
    JF_STAMP(2);

    // ---- epilogue, as gemm_f32_kernel.  C/D map: col = lane&31, row = (e&3) + 8*(e>>2) + 4*(lane>>5) ----
    float* Cout = P.C + (long long)ks * P.slab_stride;
    const bool add_bias = (P.bias != nullptr) && ks == 0;
    float local = 0.f;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn0 + j * 32 + r;
        if (n >= P.N) continue;
        const float bv = add_bias ? P.bias[n] : 0.f;
        float e_mean = 0.f, e_scale = 1.f, e_shift = 0.f;
        if (P.epi == JAMIE_EPI_BN_EVAL) {
            e_mean = P.aux0[n];
            e_scale = rsqrtf(P.aux1[n] + P.eps) * P.aux2[n];
            e_shift = P.aux3[n];
        }
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
                    if (P.store_nt) __builtin_nontemporal_store(v, cp);     // weight gradients: see gemm_bf16.hip
                    else *cp = v;
                    local += v * v;                                         // (used when `partial` is given: the clip norm)
                } else if (P.epi == JAMIE_EPI_MSE) {
                    const float d = v - P.aux0[(long long)m * P.aux_ld + n];
                    local += d * d;
                    *cp = d * P.scale;
                } else {  // JAMIE_EPI_BN_EVAL
                    const float y = (v - e_mean) * e_scale + e_shift;
                    *cp = y > 0.f ? y : P.slope * y;
                }
            }
        }
    }
    if (P.epi == JAMIE_EPI_MSE && P.partial != nullptr) {
        const float tot = block_sum(local, red);
        if (tid == 0) P.partial[t] = tot * P.pscale;
    }
    // weight gradients: the tile's sum of squares (fixed order) for the global-norm clip, as in the bf16 kernel -- the
    // separate pass over the 161 MB gradient buffer (27 us) is then only a pass over the ranges no GEMM writes
    if (P.epi == JAMIE_EPI_STORE && P.partial != nullptr) {
        const float tot = block_sum(local, red);
        if (tid == 0) P.partial[t] = tot;
    }
#ifdef JAMIE_GEMMB_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    JF_STAMP(3);
#endif
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
// `solo_lds` > 0: that many bytes of dynamic LDS on top of the kernel's own, so that only ONE workgroup fits a CU (configuration
// 19: the forward launches that run beside the optimiser stream leave half of each CU's wave slots to clip + Adam -- one workgroup
// alone on a CU keeps 0.86 of the pair's MFMA rate, profiles/r04_stamps_f32_launches_after.log)
template <int BM, int BN, int BK, int WM, int WN, bool A_KC, bool B_KC, bool MID = false, int X3 = 0>
static int launch_cfg(const jamie_gemm_problem* pr, int count, hipStream_t st, int solo_lds = 0) {
    GemmGroup g;
    memset(&g, 0, sizeof(g));
    g.count = count;
    int tiles = 0;
    bool fast = true, big = true, tails = false;
    for (int i = 0; i < count; ++i) {
        const jamie_gemm_problem& s = pr[i];
        GemmDev& d = g.p[i];
        d.A = s.A; d.B = s.B; d.C = s.C; d.bias = s.bias;
        d.aux0 = s.aux0; d.aux1 = s.aux1; d.aux2 = s.aux2; d.aux3 = s.aux3;
        d.partial = s.partial; d.a_rows = s.a_rows;
        d.slab_stride = s.slab_stride;
        d.M = s.M; d.N = s.N; d.K = s.K; d.lda = s.lda; d.ldb = s.ldb; d.ldc = s.ldc; d.aux_ld = s.aux_ld;
        d.splitk = s.splitk < 1 ? 1 : s.splitk;
        int kc = (s.K + d.splitk - 1) / d.splitk;
        kc = ((kc + BK - 1) / BK) * BK;
        if (kc < BK) kc = BK;
        if ((A_KC ? s.K : s.M) % 4 != 0 || (B_KC ? s.K : s.N) % 4 != 0) tails = true;
        d.kchunk = kc;
        d.tiles_m = (s.M + BM - 1) / BM;
        d.tiles_n = (s.N + BN - 1) / BN;
        d.tile_begin = tiles;
        d.epi = s.epi; d.accumulate = s.accumulate; d.store_nt = s.store_nt;
        d.a_vec = ((s.lda % 4) == 0 && ((uintptr_t)s.A % 16) == 0) ? 1 : 0;
        d.b_vec = ((s.ldb % 4) == 0 && ((uintptr_t)s.B % 16) == 0) ? 1 : 0;
        d.scale = s.scale; d.slope = s.slope; d.eps = s.eps; d.pscale = s.pscale;
        d.n_tiles = d.tiles_m * d.tiles_n * d.splitk;
        g.ntiles[i] = d.n_tiles;
        const bool small = (long long)d.tiles_m * d.tiles_n * d.splitk < 65536;
        d.inv_tm = small ? (unsigned)(0x100000000ull / (unsigned)d.tiles_m) + 1u : 0u;
        d.inv_tn = small ? (unsigned)(0x100000000ull / (unsigned)d.tiles_n) + 1u : 0u;
        tiles += d.n_tiles;
        // operand extents in bytes, last row rounded up to a whole float4 (stays inside the ld-strided storage)
        const long long a_rows_n = A_KC ? s.M : s.K, a_cols = A_KC ? s.K : s.M;
        const long long b_rows_n = B_KC ? s.N : s.K, b_cols = B_KC ? s.K : s.N;
        const long long ab = ((a_rows_n - 1) * s.lda + (a_cols + 3) / 4 * 4) * 4;
        const long long bb = ((b_rows_n - 1) * s.ldb + (b_cols + 3) / 4 * 4) * 4;
        if (!d.a_vec || !d.b_vec || s.a_rows || ab >= 0xFFFFFFF0LL || bb >= 0xFFFFFFF0LL) fast = false;
        d.a_bytes = (unsigned)ab; d.b_bytes = (unsigned)bb;
        const long long cb = ((long long)(s.M - 1) * s.ldc + s.N) * 4;
        d.c_bytes = cb < 0xFFFFFFF0LL ? (unsigned)cb : 0u;
        if (s.M <= 64 || s.N <= 64 || s.K <= 64) big = false;
    }
    if (tiles == 0) return 0;
    if constexpr (X3 != 0) {
        // bf16x3: the buffer-descriptor instances only (every layer of every BASELINE configuration); anything else takes the fp32
        // pipe on configuration 17's tile (same BM x BN: per-tile partial buffers sized for one fit the other)
        if (!(fast && !tails)) {
            if constexpr (BM == 128) return launch_cfg<128, 128, 32, 4, 4, A_KC, B_KC, true>(pr, count, st);
            else return launch_cfg<256, 128, 32, 4, 4, A_KC, B_KC>(pr, count, st);
        }
        if (big) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BK, WM, WN, A_KC, B_KC, 2, 1, false, X3>), dim3(tiles), dim3(WM * WN * 64), 0, st, g);
        else hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BK, WM, WN, A_KC, B_KC, 2, 0, false, X3>), dim3(tiles), dim3(WM * WN * 64), 0, st, g);
        return jamie_launch_status("jamie_gemm_f32");
    }
    if (fast && big && !tails) {
        auto kern = gemm_f32_kernel<BM, BN, BK, WM, WN, A_KC, B_KC, 2, 1, MID>;
        if (solo_lds > 0) {
            static bool raised = false;          // (once per process: the dynamic-LDS ceiling of this instantiation)
            if (!raised) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, solo_lds);
                if (e != hipSuccess) return jamie_fail((int)e, "%s: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed [%lld %lld]", "jamie_gemm_f32", solo_lds, 0);
                raised = true;
            }
        }
        hipLaunchKernelGGL(kern, dim3(tiles), dim3(WM * WN * 64), solo_lds > 0 ? solo_lds : 0, st, g);
    } else if (fast && !tails)
        hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BK, WM, WN, A_KC, B_KC, 2, 0, MID>), dim3(tiles),
                           dim3(WM * WN * 64), 0, st, g);
    else if (fast)
        hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BK, WM, WN, A_KC, B_KC, 1, 0, MID>), dim3(tiles),
                           dim3(WM * WN * 64), 0, st, g);
    else
        hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BK, WM, WN, A_KC, B_KC, 0, 0, false>), dim3(tiles),
                           dim3(WM * WN * 64), 0, st, g);
    return jamie_launch_status("jamie_gemm_f32");
}

// NT launches through the LDS-DMA kernel when every problem qualifies (16-byte aligned operands, lda / ldb multiples
// of 4, no row gather, < 4 GiB); otherwise the register-staged kernel with the same 64x64 tile.
template <int NB, int BM = 64, int BN = 64, int WM = 2, int WN = 2>
static int launch_dma_nt(const jamie_gemm_problem* pr, int count, hipStream_t st) {
    constexpr int BK = 32;
    GemmGroup g;
    memset(&g, 0, sizeof(g));
    g.count = count;
    int tiles = 0;
    bool ok = true, big = true;
    for (int i = 0; i < count; ++i) {
        const jamie_gemm_problem& s = pr[i];
        GemmDev& d = g.p[i];
        d.A = s.A; d.B = s.B; d.C = s.C; d.bias = s.bias;
        d.aux0 = s.aux0; d.aux1 = s.aux1; d.aux2 = s.aux2; d.aux3 = s.aux3;
        d.partial = s.partial; d.a_rows = s.a_rows;
        d.slab_stride = s.slab_stride;
        d.M = s.M; d.N = s.N; d.K = s.K; d.lda = s.lda; d.ldb = s.ldb; d.ldc = s.ldc; d.aux_ld = s.aux_ld;
        d.splitk = s.splitk < 1 ? 1 : s.splitk;
        int kc = (s.K + d.splitk - 1) / d.splitk;
        kc = ((kc + BK - 1) / BK) * BK;
        d.kchunk = kc;
        d.tiles_m = (s.M + BM - 1) / BM;
        d.tiles_n = (s.N + BN - 1) / BN;
        d.tile_begin = tiles;
        d.epi = s.epi; d.accumulate = s.accumulate; d.store_nt = s.store_nt;
        d.scale = s.scale; d.slope = s.slope; d.eps = s.eps; d.pscale = s.pscale;
        d.n_tiles = d.tiles_m * d.tiles_n * d.splitk;
        g.ntiles[i] = d.n_tiles;
        const bool small = (long long)d.tiles_m * d.tiles_n * d.splitk < 65536;
        d.inv_tm = small ? (unsigned)(0x100000000ull / (unsigned)d.tiles_m) + 1u : 0u;
        d.inv_tn = small ? (unsigned)(0x100000000ull / (unsigned)d.tiles_n) + 1u : 0u;
        tiles += d.n_tiles;
        if ((s.lda % 4) || (s.ldb % 4) || ((uintptr_t)s.A % 16) || ((uintptr_t)s.B % 16) || s.a_rows ||
            ((long long)(s.M - 1) * s.lda + s.K) * 4 >= 0xFFFFFFF0LL || ((long long)(s.N - 1) * s.ldb + s.K) * 4 >= 0xFFFFFFF0LL)
            ok = false;
        if (s.M <= 64 || s.N <= 64 || s.K <= 64) big = false;
    }
    if (!ok) return launch_cfg<BM, BN, 32, WM, WN, true, true>(pr, count, st);
    if (tiles == 0) return 0;
    if (big)
        hipLaunchKernelGGL((gemm_f32_dma_kernel<BM, BN, WM, WN, 1, NB>), dim3(tiles), dim3(WM * WN * 64), 0, st, g);
    else
        hipLaunchKernelGGL((gemm_f32_dma_kernel<BM, BN, WM, WN, 0, NB>), dim3(tiles), dim3(WM * WN * 64), 0, st, g);
    return jamie_launch_status("jamie_gemm_f32");
}

// Tile configurations (BM, BN, BK, WM, WN).  cfg < 0 selects by shape (see pick_cfg).
//   0: 64x128x32, 4 waves of 32x64   1: 64x64x32, 4 waves of 32x32    2: 128x128x16, 4 waves of 64x64
//   3: 128x64x32, 4 waves of 64x32   4: 128x128x32, 8 waves of 64x32  5: 32x128x32, 4 waves of 32x32
template <bool A_KC, bool B_KC>
static int launch_layout(const jamie_gemm_problem* pr, int count, int cfg, hipStream_t st) {
    switch (cfg) {
        case 0: return launch_cfg<64, 128, 32, 2, 2, A_KC, B_KC>(pr, count, st);
        case 1: return launch_cfg<64, 64, 32, 2, 2, A_KC, B_KC>(pr, count, st);
        case 2: return launch_cfg<128, 128, 16, 2, 2, A_KC, B_KC>(pr, count, st);
        case 3: return launch_cfg<128, 64, 32, 2, 2, A_KC, B_KC>(pr, count, st);
        case 4: return launch_cfg<128, 128, 32, 2, 4, A_KC, B_KC>(pr, count, st);
        case 5: return launch_cfg<32, 128, 32, 1, 4, A_KC, B_KC>(pr, count, st);
        case 6: return launch_cfg<64, 64, 64, 2, 2, A_KC, B_KC>(pr, count, st);
        case 10: return launch_cfg<64, 128, 32, 2, 4, A_KC, B_KC>(pr, count, st);     // 8 waves of 32x32
        case 11: return launch_cfg<128, 64, 32, 4, 2, A_KC, B_KC>(pr, count, st);     // 8 waves of 32x32
        case 12: return launch_cfg<128, 128, 32, 4, 4, A_KC, B_KC>(pr, count, st);    // 16 waves of 32x32
        case 7: case 8: case 9:      // 64x64x32 LDS-DMA, 3 / 2 / 4 buffers (NT only; other layouts: the register-staged 64x64)
            if constexpr (A_KC && B_KC) {
                return cfg == 7 ? launch_dma_nt<3>(pr, count, st) : cfg == 8 ? launch_dma_nt<2>(pr, count, st) : launch_dma_nt<4>(pr, count, st);
            } else {
                return launch_cfg<64, 64, 32, 2, 2, A_KC, B_KC>(pr, count, st);
            }
        case 13: case 14:      // 128x128x32 LDS-DMA on 16 waves, 2 / 3 buffers (NT only; other layouts: configuration 12)
            if constexpr (A_KC && B_KC) {
                return cfg == 13 ? launch_dma_nt<2, 128, 128, 4, 4>(pr, count, st) : launch_dma_nt<3, 128, 128, 4, 4>(pr, count, st);
            } else {
                return launch_cfg<128, 128, 32, 4, 4, A_KC, B_KC>(pr, count, st);
            }
        case 15: return launch_cfg<256, 128, 32, 4, 4, A_KC, B_KC>(pr, count, st);    // 16 waves of 64x32, one workgroup per CU
        case 16: return launch_cfg<128, 256, 32, 4, 4, A_KC, B_KC>(pr, count, st);    // 16 waves of 32x64
        case 17: return launch_cfg<128, 128, 32, 4, 4, A_KC, B_KC, true>(pr, count, st);   // 12 with the barrier in mid k-step
        case 18: return launch_cfg<64, 64, 32, 2, 2, A_KC, B_KC, true>(pr, count, st);      // 1 likewise
        // 17 with 56 KB of unused dynamic LDS: one workgroup per CU (64 + 56 KB each of 160), for launches beside the optimiser stream
        case 19: return launch_cfg<128, 128, 32, 4, 4, A_KC, B_KC, true>(pr, count, st, 56 * 1024);
        // the products on the bf16 matrix pipe, every fp32 element as three bf16 pieces (see the kernel): four waves of 64 x 64
        case 20: return launch_cfg<128, 128, 32, 2, 2, A_KC, B_KC, false, 1>(pr, count, st);
        case 21: return launch_cfg<256, 128, 32, 2, 2, A_KC, B_KC, false, 1>(pr, count, st);      // ... four waves of 128 x 64, two LDS stages
        default: return jamie_fail(-1, "%s: unknown tile configuration [%lld %lld]", "jamie_gemm_f32", cfg, 0);
    }
}

static int pick_cfg(int layout, int max_m, int max_n, int max_k) {
    // small-N or small-K problems (heads, latent -> hidden) take the 64x64 tile: more workgroups, and the
    // larger instances then only ever run the big d <-> 2d Linear layers (clean per-kernel profiles)
    // measured on MI355X at the config-2 layer shapes (tools/bench_gemm.py, profiles/): the 64x64x32 tile
    // (3-4 workgroups per CU, finest load balance between the two modalities) wins or ties every layout
    (void)layout; (void)max_m; (void)max_n; (void)max_k;
    return 1;         // (18, the same tile with the barrier in mid k-step: no difference on these short launches)
}

extern "C" int jamie_gemm_f32_cfg(const jamie_gemm_problem* pr, int count, int layout, int cfg, void* stream) {
    JAMIE_ARG(pr != nullptr && count >= 1 && count <= JAMIE_MAX_GEMM_GROUP_F32, "1 <= count <= JAMIE_MAX_GEMM_GROUP_F32");
    JAMIE_ARG(layout >= JAMIE_NT && layout <= JAMIE_TN, "layout");
    int max_n = 0, max_m = 0, max_k = 0;
    for (int i = 0; i < count; ++i) {
        const jamie_gemm_problem& s = pr[i];
        JAMIE_ARG(s.A && s.B && s.C, "null operand");
        JAMIE_ARG(s.M > 0 && s.N > 0 && s.K > 0, "empty problem");
        JAMIE_ARG(!s.c_panel, "c_panel: jamie_gemm_bf16's large-tile configurations only");
        JAMIE_ARG(s.ldc >= s.N, "ldc < N");
        if (layout == JAMIE_NT) JAMIE_ARG(s.lda >= s.K && s.ldb >= s.K, "NT: lda/ldb < K");
        if (layout == JAMIE_NN) JAMIE_ARG(s.lda >= s.K && s.ldb >= s.N, "NN: lda < K or ldb < N");
        if (layout == JAMIE_TN) JAMIE_ARG(s.lda >= s.M && s.ldb >= s.N, "TN: lda < M or ldb < N");
        JAMIE_ARG(s.epi >= JAMIE_EPI_STORE && s.epi <= JAMIE_EPI_BN_EVAL, "epilogue id");
        JAMIE_ARG(s.epi == JAMIE_EPI_STORE || s.splitk <= 1, "fused epilogues need splitk == 1");
        JAMIE_ARG(s.epi != JAMIE_EPI_STORE || !s.partial || s.splitk <= 1, "a sum-of-squares partial needs splitk == 1");
        JAMIE_ARG(s.splitk <= 1 || !s.accumulate, "split-K slabs cannot accumulate");
        JAMIE_ARG(s.epi != JAMIE_EPI_MSE || (s.aux0 && s.aux_ld >= s.N), "MSE epilogue needs aux0 = X");
        JAMIE_ARG(s.epi != JAMIE_EPI_BN_EVAL || (s.aux0 && s.aux1 && s.aux2 && s.aux3), "BN_EVAL needs aux0..3");
        JAMIE_ARG(layout != JAMIE_TN || s.a_rows == nullptr, "a_rows only for NT/NN");
        JAMIE_ARG(!s.c_bf16 && !s.b_tr && !s.a_tr, "c_bf16 / a_tr / b_tr belong to jamie_gemm_bf16");
        JAMIE_ARG(s.splitk <= 1 || s.slab_stride >= (long long)s.M * s.ldc, "slab_stride too small");
        if (s.N > max_n) max_n = s.N;
        if (s.M > max_m) max_m = s.M;
        if (s.K > max_k) max_k = s.K;
    }
    hipStream_t st = (hipStream_t)stream;
    if (cfg < 0) cfg = pick_cfg(layout, max_m, max_n, max_k);
    if (layout == JAMIE_NT) return launch_layout<true, true>(pr, count, cfg, st);
    if (layout == JAMIE_NN) return launch_layout<true, false>(pr, count, cfg, st);
    return launch_layout<false, false>(pr, count, cfg, st);
}

extern "C" int jamie_gemm_f32(const jamie_gemm_problem* pr, int count, int layout, void* stream) {
    return jamie_gemm_f32_cfg(pr, count, layout, -1, stream);
}

// tile geometry of a configuration (host helper: sizing of per-tile partial buffers)
extern "C" int jamie_gemm_tile(int layout, int max_m, int max_n, int max_k, int cfg, int* bm, int* bn) {
    static const int T[22][2] = {{64, 128}, {64, 64}, {128, 128}, {128, 64}, {128, 128}, {32, 128}, {64, 64}, {64, 64}, {64, 64}, {64, 64},
                                 {64, 128}, {128, 64}, {128, 128}, {128, 128}, {128, 128}, {256, 128}, {128, 256}, {128, 128}, {64, 64}, {128, 128},
                                 {128, 128}, {256, 128}};
    if (cfg < 0) cfg = pick_cfg(layout, max_m, max_n, max_k);
    if (cfg > 21 || !bm || !bn) return jamie_fail(-1, "%s: bad arguments [%lld %lld]", "jamie_gemm_tile", cfg, 0);
    *bm = T[cfg][0]; *bn = T[cfg][1];
    return 0;
}

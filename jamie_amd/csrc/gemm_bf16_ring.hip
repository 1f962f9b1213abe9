// Persistent loader / consumer ring GEMM for the BACKWARD products of JAMIE's Linear layers in bf16 compute mode, gfx950
// (reference: autograd of model.py:151,161,192,197,207 inside jamie.py:734 `batch_loss.backward()`):
//     dX = dy W       A = dy [B, out] (K contiguous)          B = W [out, in] = [K, N] as stored       (b_tr)
//     dW = dy^T a     A = dy [B, out] = [K, M] as stored      B = a [B, in]  = [K, N] as stored  (a_tr + b_tr)
// grouped in one launch per layer (jamie_gemm_bf16 runs the same products as one workgroup per tile).
//
// Why another kernel (profiles/r04_stamps_probe*.log, tools/stamp_gemm_bf16.py): in the one-tile-per-workgroup kernel a
// workgroup retires one 128 x 128 x 64 k-step (32 KB of operands) per ~1.0-1.2 us WHATEVER its ring depth (2 .. 5 buffers:
// 1.06-1.09 us for the dX tiles) and WHATEVER the cache state of its operands (hot: 1.01-1.06 us): every wave issues its
// LDS-DMA pieces itself, in a burst behind the k-step barrier, and stalls at the issue while the matrix pipe idles; two
// workgroups per CU get 55 GB/s per CU out of that.  On top of it a dW tile (K = batch = 8 k-steps) spends 3.9 us waiting for
// its first operands and 4.6 us storing for 9 us of k-loop, and the long dX tiles (32 k-steps) of a layer end 10-15 us after
// the rest of the grid.  Here (MI355X_MICROARCH.md row ring-gemm; cdna_hip_programming.md 5.6):
//   * ONE persistent workgroup per CU: 4 LOADER waves stream the operands of a static tile list into a ring of 32 KB slots by
//     LDS-DMA (`buffer_load_dwordx4 ... lds`: out-of-range k-rows of the last, partial k-step read as zero) and never touch
//     the matrix pipe; 8 CONSUMER waves (64 x 32 of the tile each, v_mfma_f32_32x32x16_bf16) never issue a load.  No
//     s_barrier after start-up: per slot a FULL word per loader wave and a FREE word per consumer wave in LDS (generation
//     numbers, never reset).
//   * the ring runs ACROSS tiles: the next tile's first k-steps land while the consumers store the current one, so a tile
//     costs its k-steps and its store instructions, not a memory round trip at either end;
//   * the tile lists are balanced on the host by k-steps (longest first within an XCD's share of every problem, so the
//     panels a tile shares with its neighbours stay in one L2).
// Results are bit-identical to jamie_gemm_bf16 (same MFMA order inside a tile; the partial k-step adds exact zeros).
// (experiments build only: -DJAMIE_EXPERIMENTS, jamie_amd.build.build_experiments(); not in the product library)
#ifdef JAMIE_EXPERIMENTS
#include "common.h"
#include "range_norm.h"
#include <algorithm>
#include <type_traits>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

// diagnostic ablations (timing only, wrong results; A/B builds with -DRG_ABL=n): 1 = the consumers neither read fragments nor
// multiply (what the loaders can deliver), 2 = the loaders issue no DMA (what the consumers can take)
#ifndef RG_ABL
#define RG_ABL 0
#endif
#ifndef RG_NB
#define RG_NB 4
#endif
#ifndef RG_LA
#define RG_LA 2
#endif
#ifndef RG_NLOAD
#define RG_NLOAD 4         // loader waves (4 or 8): a wave's LDS-DMA instructions retire at ~1 per 100 ns on these operands
#endif                     // (4 loader waves alone deliver 36 GB/s per CU: profiles/r04_ring_ablations_v1.log)
#define RG_OOB 0xFFFFFFF0u
#define RG_MAX_ITEMS 48          // tiles per workgroup (a launch of more falls back to jamie_gemm_bf16)
#define RG_SPIN_MAX (1u << 22)   // every poll is bounded: a broken hand-off ends the launch with the error word set, never hangs

struct RingDev {
    const unsigned short* A; const unsigned short* B; void* C; float* partial;
    long long slab_stride;       // elements of C between split-K slabs
    int M, N, K, lda, ldb, ldc;
    int splitk, kchunk, tiles_m, tiles_n;
    int a_tr, store_nt, c_bf16;
    unsigned a_bytes, b_bytes, c_bytes;
    float pscale;
};
struct RingGroup { RingDev p[JAMIE_MAX_GEMM_GROUP]; int count; };

#ifdef JAMIE_GEMMB_STAMP
__device__ unsigned long long jamie_dbg_ring_stamps[512 * 64];
extern "C" int jamie_debug_ring_stamps(unsigned long long* host_out, int n_blocks) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(jamie_dbg_ring_stamps), sizeof(unsigned long long) * 64 * n_blocks);
}
#define RG_STAMP(k) do { if ((threadIdx.x & 63) == 0 && wid == NLOAD && blockIdx.x < 512 && (k) < 64) jamie_dbg_ring_stamps[blockIdx.x * 64 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define RG_STAMPL(k) do { if ((threadIdx.x & 63) == 0 && wid == 0 && blockIdx.x < 512 && (k) < 64) jamie_dbg_ring_stamps[blockIdx.x * 64 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define RG_STAMP(k) do {} while (0)
#define RG_STAMPL(k) do {} while (0)
#endif

typedef void __attribute__((address_space(3)))* rg_lptr_t;

template <int N>
__device__ __forceinline__ void rg_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// Hand-off words in LDS.  Consumer waves (no LDS-DMA anywhere in their code path) use relaxed workgroup-scope atomics: plain
// ds_read / ds_write that the optimiser neither hoists out of a poll loop nor fences (a `volatile` access makes hipcc wait
// vmcnt(0) around it).  LOADER waves use inline asm: hipcc's wait-count pass puts `s_waitcnt vmcnt(0)` in front of every LDS
// access it can see while an LDS-DMA is pending (it cannot prove that the access does not read what the DMA writes), which would
// drain the loader's queue at every poll / publish; the asm is invisible to that pass.
__device__ __forceinline__ int rg_ld(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void rg_st(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ unsigned rg_lds_addr(const void* p) {
    typedef const void __attribute__((address_space(3)))* lp_t;
    return (unsigned)reinterpret_cast<size_t>((lp_t)p);
}
__device__ __forceinline__ int rg_min8_asm(const int* p) {          // min of 8 consecutive words (32-byte aligned)
    u32x4 a, b;
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(a), "=&v"(b) : "v"(rg_lds_addr(p)) : "memory");
    const int m0 = min(min((int)a.x, (int)a.y), min((int)a.z, (int)a.w));
    const int m1 = min(min((int)b.x, (int)b.y), min((int)b.z, (int)b.w));
    return min(m0, m1);
}
__device__ __forceinline__ int rg_ld_asm(const int* p) {
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(rg_lds_addr(p)) : "memory");
    return v;
}
__device__ __forceinline__ void rg_st_asm(int* p, int v) {
    asm volatile("ds_write_b32 %0, %1" ::"v"(rg_lds_addr(p)), "v"(v) : "memory");
}

// NB ring slots of 32 KB; LA = fills a loader wave leaves in flight behind the one it publishes
template <int NB, int LA>
__global__ __launch_bounds__((RG_NLOAD + 8) * 64) void gemm_bf16_ring_kernel(RingGroup g, const int* __restrict__ sched, int max_items, RangeRide rr,
                                                             int ride_blocks, unsigned* __restrict__ err) {
    constexpr int BM = 128, BN = 128, BK = 64;
    constexpr int NLOAD = RG_NLOAD, NCONS = 8, WN = 4, NT = (NLOAD + NCONS) * 64;
    constexpr int A_SZ = BM * 128, B_SZ = BN * 128, T_SZ = A_SZ + B_SZ;      // 32 KB per slot
    constexpr int PL = 16 / NLOAD;                          // 1-KiB pieces per loader wave and operand (16 per operand)
    constexpr int GLD = 2 * PL;                             // LDS-DMA instructions per loader wave and fill
    constexpr int SROW = 36, SCR_W = 16 * SROW * 4;         // epilogue scratch: 16 rows x (32 + 4 pad) floats per consumer wave
    constexpr int SETS = NB + 1;                            // reduction slots (consumer waves drift apart by < NB k-steps)
    constexpr int CTRL = NB * T_SZ + NCONS * SCR_W;
    static_assert(LA >= 1 && LA < NB && GLD * LA <= 63, "loader depth");
    static_assert(NLOAD == 4 || NLOAD == 8, "loader waves");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[CTRL + 4 * (RG_MAX_ITEMS + NB * 8 + NB * 8 + 2 * SETS * 8 + 16) + 2048];
    int* items = reinterpret_cast<int*>(smem + CTRL);
    int* fullw = items + RG_MAX_ITEMS;                      // [NB][8]  generation published by loader wave l (NLOAD of the 8 words used)
    int* freew = fullw + NB * 8;                            // [NB][8]  generation released by consumer wave c
    float* redv = reinterpret_cast<float*>(freew + NB * 8); // [SETS][8]
    int* redt = reinterpret_cast<int*>(redv + SETS * 8);    // [SETS][8]
    float* ride_red = reinterpret_cast<float*>(redt + SETS * 8 + 16);

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int w = blockIdx.x;
    // ---- start-up (the only workgroup barriers of the launch) ----
    for (int i = tid; i < RG_MAX_ITEMS; i += NT) items[i] = i < max_items ? sched[(long long)w * max_items + i] : -1;
    for (int i = tid; i < NB * 8 + NB * 8 + 2 * SETS * 8; i += NT) fullw[i] = 0;
    __syncthreads();
    int n_items = 0, total_k = 0;
    for (int i = 0; i < RG_MAX_ITEMS; ++i) {
        const int it = items[i];
        if (it < 0) break;
        const RingDev& P = g.p[__builtin_amdgcn_readfirstlane(it >> 24)];
        const int ks = (it & 0xFFFFFF) / (P.tiles_m * P.tiles_n);
        const int kbeg = ks * P.kchunk, kend = min(P.K, kbeg + P.kchunk);
        total_k += (kend - kbeg + BK - 1) / BK;
        ++n_items;
    }
    n_items = __builtin_amdgcn_readfirstlane(n_items);
    total_k = __builtin_amdgcn_readfirstlane(total_k);
    RG_STAMPL(0);

    // the range-norm riders (jamie_gemm_bf16_ranges): one chunk per workgroup, before the roles split; every thread takes part
    if (w < ride_blocks) {
        range_ride_block(rr, w, ride_red);
        __syncthreads();
    }

    if (wid < NLOAD) {
        // =================================== LOADER WAVES ===================================
        const int l = wid;
        int it_i = 0, kt = 0, nk = 0, cur_pi = 0;
        unsigned a_off[PL], b_off[PL], a_inc = 0, b_inc = 0;
        __amdgpu_buffer_rsrc_t a_rs, b_rs;
        auto setup = [&](int i) {
            const int it = __builtin_amdgcn_readfirstlane(rg_ld_asm(items + i));
            cur_pi = __builtin_amdgcn_readfirstlane(it >> 24);
            const RingDev& P = g.p[cur_pi];
            const int t = it & 0xFFFFFF;
            const int tm_i = t % P.tiles_m, tn_i = (t / P.tiles_m) % P.tiles_n, ks = t / (P.tiles_m * P.tiles_n);
            const int m0 = tm_i * BM, n0 = tn_i * BN;
            const int kbeg = ks * P.kchunk, kend = min(P.K, kbeg + P.kchunk);
            nk = (kend - kbeg + BK - 1) / BK;
            a_rs = __builtin_amdgcn_make_buffer_rsrc((void*)P.A, 0, (int)P.a_bytes, 0x00020000);
            b_rs = __builtin_amdgcn_make_buffer_rsrc((void*)P.B, 0, (int)P.b_bytes, 0x00020000);
#pragma unroll
            for (int q = 0; q < PL; ++q) {
                const int piece = l + NLOAD * q;
                if (P.a_tr) {      // A stored [K, M]: [64 k][128 m] image, 256-byte rows, one piece = 4 k-rows
                    const int krow = 4 * piece + (lane >> 4);
                    const int lc = (lane & 15) ^ (((krow & 3) << 2) | ((krow >> 2) & 3));
                    a_off[q] = (unsigned)(((long long)(kbeg + krow) * P.lda + min(m0 + lc * 8, P.M - 8)) * 2);
                } else {           // A stored [M, K]: [128 m][64 k] image, 128-byte rows, one piece = 8 rows
                    const int row = 8 * piece + (lane >> 3), pch = lane & 7;
                    a_off[q] = (unsigned)(((long long)min(m0 + row, P.M - 1) * P.lda + kbeg + ((pch ^ ((row >> 1) & 7)) * 8)) * 2);
                }
                {                  // B stored [K, N] (always, in this kernel)
                    const int krow = 4 * piece + (lane >> 4);
                    const int lc = (lane & 15) ^ (((krow & 3) << 2) | ((krow >> 2) & 3));
                    b_off[q] = (unsigned)(((long long)(kbeg + krow) * P.ldb + min(n0 + lc * 8, P.N - 8)) * 2);
                }
            }
            a_inc = P.a_tr ? (unsigned)(BK * P.lda * 2) : (unsigned)(BK * 2);
            b_inc = (unsigned)(BK * P.ldb * 2);
        };
        auto poll_free = [&](int slot, int gen) -> bool {      // every consumer wave has released generation `gen` of the slot
            const int* f = freew + slot * 8;
            unsigned spins = 0;
            while (true) {
                const int mn = rg_min8_asm(f);
                if (mn >= gen) return true;
                __builtin_amdgcn_s_sleep(2);
                if (++spins > RG_SPIN_MAX) { if (lane == 0) atomicOr(err, 1u); return false; }
            }
        };
        auto publish = [&](int j) {                            // fill j of this wave has landed: tell the consumers
            if (lane == 0) rg_st_asm(fullw + (j % NB) * 8 + l, j / NB + 1);
        };
        if (n_items > 0) setup(0);
        bool ok = true;
        for (int j = 0; j < total_k && ok; ++j) {
            const int slot = j % NB, gen = j / NB;
            if (gen > 0) ok = poll_free(slot, gen);
            if (!ok) break;
            unsigned char* As = smem + slot * T_SZ;
            unsigned char* Bs = As + A_SZ;
            const unsigned ka = (unsigned)kt * a_inc, kb = (unsigned)kt * b_inc;
#if RG_ABL != 2 && defined(RG_GLOBAL)
            // (timing experiment only: global_load_lds instead of the bounds-checked buffer form; partial k-steps read garbage)
            typedef const void __attribute__((address_space(1)))* gptr_t;
#pragma unroll
            for (int q = 0; q < PL; ++q)
                __builtin_amdgcn_global_load_lds((gptr_t)((const char*)g.p[cur_pi].A + (a_off[q] + ka)), (rg_lptr_t)(As + (l + NLOAD * q) * 1024), 16, 0, 0);
#pragma unroll
            for (int q = 0; q < PL; ++q)
                __builtin_amdgcn_global_load_lds((gptr_t)((const char*)g.p[cur_pi].B + min(b_off[q] + kb, g.p[cur_pi].b_bytes - 16u)), (rg_lptr_t)(Bs + (l + NLOAD * q) * 1024), 16, 0, 0);
#elif RG_ABL != 2
#pragma unroll
            for (int q = 0; q < PL; ++q)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rs, (rg_lptr_t)(As + (l + NLOAD * q) * 1024), 16, (int)(a_off[q] + ka), 0, 0, 0);
#pragma unroll
            for (int q = 0; q < PL; ++q)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rs, (rg_lptr_t)(Bs + (l + NLOAD * q) * 1024), 16, (int)(b_off[q] + kb), 0, 0, 0);
#else
            asm volatile("" ::"v"(ka), "v"(kb), "v"(a_off[0]), "v"(b_off[0]));
#endif
            if (j >= LA) {                                     // fill j - LA has landed once at most LA fills are outstanding
                rg_wait_vm<GLD * LA>();
                publish(j - LA);
            }
            if (++kt == nk) {
                kt = 0;
                if (++it_i < n_items) setup(it_i);
            }
        }
        // drain: the last LA fills, oldest first
        if (ok) {
#pragma unroll
            for (int r = LA - 1; r >= 0; --r) {
                const int j = total_k - 1 - r;
                if (j >= 0) {
                    if (r == 0) rg_wait_vm<0>();
                    else if (r == 1) rg_wait_vm<GLD>();
                    else if (r == 2) rg_wait_vm<2 * GLD>();
                    else rg_wait_vm<3 * GLD>();
                    publish(j);
                }
            }
        }
        RG_STAMPL(63);
        return;
    }

    // =================================== CONSUMER WAVES ===================================
    const int c = wid - NLOAD;
    const int wm0 = (c / WN) * 64, wn0 = (c % WN) * 32;
    const int r = lane & 31, h = lane >> 5;
    const int swz = (r >> 1) & 7;
    const int tq = (lane >> 2) & 3, tp = lane & 3, tc0 = wn0 + 16 * ((lane >> 4) & 1), ta0 = wm0 + 16 * ((lane >> 4) & 1);
    typedef s16x4 __attribute__((address_space(3)))* trp_t;
    bf16x8 af[2][2], bf[2];
    // fragments of sub-step s of the slot at `As` (k-contiguous A: one ds_read_b128; k-row-major operands: two ds_read_b64_tr_b16,
    // the image and the swizzle of gemm_bf16.hip's large-tile kernel)
    auto read_frags = [&](const unsigned char* As, int fb, int s, auto tra) {
        const unsigned char* Bs = As + A_SZ;
        if constexpr (decltype(tra)::value) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ch = ((ta0 + 32 * i) >> 3) + (tp >> 1);
                const unsigned char* base = As + s * (16 * BM * 2) + 8 * (tp & 1);
                const int o0 = (8 * h + tq) * (BM * 2) + ((ch ^ ((tq << 2) | ((2 * h) & 3))) << 4);
                const int o1 = (8 * h + 4 + tq) * (BM * 2) + ((ch ^ ((tq << 2) | ((2 * h + 1) & 3))) << 4);
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((trp_t)(base + o0));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((trp_t)(base + o1));
                af[fb][i] = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
            }
        } else {
            const int off = ((2 * s + h) ^ swz) << 4;
#pragma unroll
            for (int i = 0; i < 2; ++i)
                af[fb][i] = *reinterpret_cast<const bf16x8*>(As + (wm0 + i * 32 + r) * 128 + off);
        }
        {
            const int ch = (tc0 >> 3) + (tp >> 1);
            const unsigned char* base = Bs + s * (16 * BN * 2) + 8 * (tp & 1);
            const int o0 = (8 * h + tq) * (BN * 2) + ((ch ^ ((tq << 2) | ((2 * h) & 3))) << 4);
            const int o1 = (8 * h + 4 + tq) * (BN * 2) + ((ch ^ ((tq << 2) | ((2 * h + 1) & 3))) << 4);
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((trp_t)(base + o0));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((trp_t)(base + o1));
            bf[fb] = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
        }
    };
    auto poll_full = [&](int j) -> bool {                      // every loader wave has published fill j
        const int* f = fullw + (j % NB) * 8;
        const int want = j / NB + 1;
        unsigned spins = 0;
        while (true) {
            int mn = min(min(rg_ld(f), rg_ld(f + 1)), min(rg_ld(f + 2), rg_ld(f + 3)));
            if constexpr (NLOAD == 8) mn = min(mn, min(min(rg_ld(f + 4), rg_ld(f + 5)), min(rg_ld(f + 6), rg_ld(f + 7))));
            if (mn >= want) return true;
            __builtin_amdgcn_s_sleep(1);
            if (++spins > RG_SPIN_MAX) { if (lane == 0) atomicOr(err, 2u); return false; }
        }
    };
    auto release = [&](int j) {                                // this wave holds its last fragments of fill j in registers
        if (lane == 0) rg_st(freew + (j % NB) * 8 + c, j / NB + 1);
    };

    float* scr = reinterpret_cast<float*>(smem + NB * T_SZ + c * SCR_W);
    int j = 0;                                                 // stream index of the next k-step
    bool ok = true;
    for (int ii = 0; ii < n_items && ok; ++ii) {
        const int it = items[ii];
        const int pi = __builtin_amdgcn_readfirstlane(it >> 24);
        const RingDev& P = g.p[pi];
        const int t = it & 0xFFFFFF;
        const int tm_i = t % P.tiles_m, tn_i = (t / P.tiles_m) % P.tiles_n, ks = t / (P.tiles_m * P.tiles_n);
        const int m0 = tm_i * BM, n0 = tn_i * BN;
        const int kbeg = ks * P.kchunk, kend = min(P.K, kbeg + P.kchunk);
        const int nk = (kend - kbeg + BK - 1) / BK;
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

        auto k_loop = [&](auto tra) {
#if RG_ABL == 1
            for (int kt = 0; kt < nk && ok; ++kt, ++j) {
                ok = poll_full(j);
                release(j);
            }
            return;
#endif
            ok = poll_full(j);
            if (!ok) return;
            read_frags(smem + (j % NB) * T_SZ, 0, 0, tra);
            for (int kt = 0; kt < nk; ++kt, ++j) {
                const unsigned char* As = smem + (j % NB) * T_SZ;
#pragma unroll
                for (int s = 0; s < BK / 16; ++s) {
                    __builtin_amdgcn_sched_barrier(0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[s & 1], af[s & 1][0], acc[0], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (s + 1 < BK / 16) {
                        read_frags(As, (s + 1) & 1, s + 1, tra);
                    } else {
                        // the last fragments of this k-step are in registers (the MFMA above consumed their pair): release the
                        // slot, and fetch the next k-step's first fragments under this sub-step's second MFMA
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        release(j);
                        if (kt + 1 < nk) {
                            ok = poll_full(j + 1);
                            if (ok) read_frags(smem + ((j + 1) % NB) * T_SZ, 0, 0, tra);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[s & 1], af[s & 1][1], acc[1], 0, 0, 0);
                }
                if (!ok) { ++j; return; }
            }
        };
        if (P.a_tr) k_loop(std::true_type{}); else k_loop(std::false_type{});
        if (!ok) break;
        RG_STAMP(2 + 2 * ii);

        // ---- epilogue: transposed C/D map (m = lane & 31, n = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)) turned through a private
        // LDS scratch, 16 rows at a time, into whole 128-byte (fp32) / 64-byte (bf16) row segments; buffer stores, so that rows /
        // columns beyond the matrix are dropped by the bounds check and EVERY wave issues the same instructions ----
        const __amdgpu_buffer_rsrc_t c_rs = __builtin_amdgcn_make_buffer_rsrc(
            (void*)((char*)P.C + (long long)ks * P.slab_stride * (P.c_bf16 ? 2 : 4)), 0, (int)P.c_bytes, 0x00020000);
        const int cch = lane & 7, rsub = lane >> 3;
        const int nc = n0 + wn0 + cch * 4;
        float local = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                if ((r >> 4) == half) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<float4*>(scr + (r & 15) * SROW + 8 * q + 4 * h) =
                            make_float4(acc[i][4 * q], acc[i][4 * q + 1], acc[i][4 * q + 2], acc[i][4 * q + 3]);
                }
                // the rows are exchanged BETWEEN LANES of this wave: without a wave-level ordering point hipcc sank the read below
                // into the masked block above (only the writing lanes read; found by tools/debug_ring.py)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
                for (int rr2 = 0; rr2 < 2; ++rr2) {
                    const int row = rr2 * 8 + rsub;
                    const float4 v = *reinterpret_cast<const float4*>(scr + row * SROW + cch * 4);
                    const int m = m0 + wm0 + i * 32 + half * 16 + row;
                    const bool valid = m < P.M && nc < P.N;
                    if (valid) local += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
                    if (P.c_bf16) {
                        auto bfr = [](float x) { return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)x); };
                        u32x2 pk;
                        pk.x = bfr(v.x) | (bfr(v.y) << 16);
                        pk.y = bfr(v.z) | (bfr(v.w) << 16);
                        const unsigned off = valid ? ((unsigned)m * (unsigned)P.ldc + (unsigned)nc) * 2u : RG_OOB;
                        __builtin_amdgcn_raw_buffer_store_b64(pk, c_rs, (int)off, 0, 2);          // (weight gradients: non-temporal)
                    } else {
                        u32x4 pk;
                        pk.x = __float_as_uint(v.x); pk.y = __float_as_uint(v.y); pk.z = __float_as_uint(v.z); pk.w = __float_as_uint(v.w);
                        const unsigned off = valid ? ((unsigned)m * (unsigned)P.ldc + (unsigned)nc) * 4u : RG_OOB;
                        if (P.store_nt) __builtin_amdgcn_raw_buffer_store_b128(pk, c_rs, (int)off, 0, 2);
                        else __builtin_amdgcn_raw_buffer_store_b128(pk, c_rs, (int)off, 0, 0);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // (reads done before the next pass overwrites the rows)
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
        // per-tile sum of squares of what was stored (the clip norm's partial sums, jamie.py:739): the consumer waves' sums meet in
        // a slot of a small ring in LDS (tagged with the item, so no reset and no barrier); wave 0 adds them in wave order
        if (P.partial != nullptr) {
            const float ws = wave_sum_dpp(local);
            const int set = ii % SETS;
            if (lane == 0) {
                rg_st(reinterpret_cast<int*>(redv) + set * 8 + c, __builtin_bit_cast(int, ws));
                rg_st(redt + set * 8 + c, ii + 1);
            }
            if (c == 0) {
                const int* tg = redt + set * 8;
                unsigned spins = 0;
                while (true) {
                    int mn = rg_ld(tg);
#pragma unroll
                    for (int q = 1; q < NCONS; ++q) mn = min(mn, rg_ld(tg + q));
                    if (mn >= ii + 1) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > RG_SPIN_MAX) { if (lane == 0) atomicOr(err, 4u); ok = false; break; }
                }
                if (ok && lane == 0) {
                    const int* rv = reinterpret_cast<const int*>(redv) + set * 8;
                    float tot = 0.f;
#pragma unroll
                    for (int q = 0; q < NCONS; ++q) tot += __builtin_bit_cast(float, rg_ld(rv + q));
                    P.partial[t] = tot * P.pscale;
                }
            }
        }
        RG_STAMP(3 + 2 * ii);
    }
    RG_STAMP(62);
}

// ------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------
static int ring_fill(const jamie_gemm_problem* pr, int count, RingGroup* g, int* nk_out /*[count]*/) {
    memset(g, 0, sizeof(*g));
    g->count = count;
    for (int i = 0; i < count; ++i) {
        const jamie_gemm_problem& s = pr[i];
        JAMIE_ARG(s.A && s.B && s.C, "null operand");
        JAMIE_ARG(s.M > 0 && s.N > 0 && s.K > 0, "empty problem");
        JAMIE_ARG(s.b_tr, "the ring kernel takes backward products only: B stored [K, N] (b_tr)");
        JAMIE_ARG(s.epi == JAMIE_EPI_STORE && !s.accumulate && s.bias == nullptr && s.a_rows == nullptr, "plain store, no bias, no accumulate");
        JAMIE_ARG(s.K % 8 == 0 && s.lda % 8 == 0 && s.ldb % 8 == 0 && s.N % 8 == 0 && s.N >= 8 && s.ldb >= s.N, "K, N, lda, ldb multiples of 8");
        JAMIE_ARG(!s.a_tr || (s.M % 8 == 0 && s.M >= 8 && s.lda >= s.M), "a_tr: M a multiple of 8");
        JAMIE_ARG(s.a_tr || s.lda >= s.K, "lda");
        JAMIE_ARG(s.ldc >= s.N && s.ldc % 4 == 0 && (uintptr_t)s.C % 16 == 0 && s.slab_stride % 4 == 0, "C: 16-byte aligned rows");
        JAMIE_ARG(((uintptr_t)s.A % 16) == 0 && ((uintptr_t)s.B % 16) == 0, "bf16 operands must be 16-byte aligned");
        JAMIE_ARG(!s.c_bf16 || (s.splitk <= 1), "c_bf16: no split-K");
        JAMIE_ARG(!s.partial || s.splitk <= 1, "sum-of-squares partials need splitk == 1");
        JAMIE_ARG(s.splitk <= 1 || s.slab_stride >= (long long)s.M * s.ldc, "slab_stride too small");
        const long long a_b = (s.a_tr ? ((long long)(s.K - 1) * s.lda + s.M) : ((long long)(s.M - 1) * s.lda + s.K)) * 2;
        const long long b_b = ((long long)(s.K - 1) * s.ldb + s.N) * 2;
        const long long c_b = (long long)s.M * s.ldc * (s.c_bf16 ? 2 : 4);
        JAMIE_ARG(a_b < 0xFFFFFFF0LL && b_b < 0xFFFFFFF0LL && c_b < 0xFFFFFFF0LL, "operands must stay below 4 GiB");
        RingDev& d = g->p[i];
        d.A = (const unsigned short*)s.A; d.B = (const unsigned short*)s.B; d.C = (void*)s.C; d.partial = s.partial;
        d.slab_stride = s.slab_stride;
        d.M = s.M; d.N = s.N; d.K = s.K; d.lda = s.lda; d.ldb = s.ldb; d.ldc = s.ldc;
        d.splitk = s.splitk < 1 ? 1 : s.splitk;
        int kc = (s.K + d.splitk - 1) / d.splitk;
        d.kchunk = ((kc + 63) / 64) * 64;
        d.tiles_m = (s.M + 127) / 128;
        d.tiles_n = (s.N + 127) / 128;
        d.a_tr = s.a_tr; d.store_nt = s.store_nt; d.c_bf16 = s.c_bf16;
        d.a_bytes = (unsigned)a_b; d.b_bytes = (unsigned)b_b; d.c_bytes = (unsigned)c_b;
        d.pscale = s.pscale;
        if (nk_out) nk_out[i] = (d.kchunk + 63) / 64;
    }
    return 0;
}

// Static schedule: workgroup w (one per CU; w & 7 labels the XCD it is dealt to, MI355X_MICROARCH.md) gets a list of tiles,
// (problem << 24) | tile, terminated by -1.  Every problem's tile list (the M-tiles of one operand panel adjacent) is cut into
// 8 contiguous chunks, one per XCD label, as in jamie_gemm_bf16; inside an XCD the chunks' tiles are dealt longest first to the
// workgroup with the least k-steps so far (+ 2 k-steps of fixed cost per tile).
extern "C" int jamie_gemm_bf16_ring_plan(const jamie_gemm_problem* pr, int count, int n_wg, int max_items, int32_t* sched /*host*/) {
    JAMIE_ARG(pr != nullptr && sched != nullptr && count >= 1 && count <= JAMIE_MAX_GEMM_GROUP, "1 <= count <= JAMIE_MAX_GEMM_GROUP");
    JAMIE_ARG(n_wg >= 8 && n_wg % 8 == 0 && max_items >= 1 && max_items <= RG_MAX_ITEMS, "n_wg a multiple of 8, max_items <= 48");
    RingGroup g;
    const int rc = ring_fill(pr, count, &g, nullptr);
    if (rc) return rc;
    const int per = n_wg / 8;
    for (long long i = 0; i < (long long)n_wg * max_items; ++i) sched[i] = -1;
    int rot = 0;
    struct Tile { int nk, code; };
    std::vector<Tile> tiles[8];
    for (int i = 0; i < count; ++i) {
        const RingDev& d = g.p[i];
        const int T = d.tiles_m * d.tiles_n * d.splitk, qp = T >> 3, rp = T & 7;
        JAMIE_ARG(T < (1 << 24), "too many tiles");
        for (int x = 0; x < 8; ++x) {
            const int jj = (x - rot) & 7;
            const int cp = qp + (jj < rp ? 1 : 0), first = jj * qp + (jj < rp ? jj : rp);
            for (int q = 0; q < cp; ++q) {
                const int t = first + q;
                const int ks = t / (d.tiles_m * d.tiles_n);
                const int kbeg = ks * d.kchunk, kend = (d.K < kbeg + d.kchunk) ? d.K : kbeg + d.kchunk;
                tiles[x].push_back(Tile{(kend - kbeg + 63) / 64, (i << 24) | t});
            }
        }
        rot = (rot + rp) & 7;
    }
    JAMIE_ARG(per <= 512, "too many workgroups per XCD");
    for (int x = 0; x < 8; ++x) {
        std::vector<Tile>& tl = tiles[x];
        std::stable_sort(tl.begin(), tl.end(), [](const Tile& a, const Tile& b) { return a.nk > b.nk; });      // longest first
        int load[512], cnt[512];
        for (int s = 0; s < per; ++s) { load[s] = 0; cnt[s] = 0; }
        for (size_t a = 0; a < tl.size(); ++a) {
            int best = 0;
            for (int s = 1; s < per; ++s) if (load[s] < load[best]) best = s;
            if (cnt[best] >= max_items) return jamie_fail(-1, "%s: more than max_items tiles per workgroup [%lld %lld]", "jamie_gemm_bf16_ring_plan", cnt[best], max_items);
            const int wg = x + 8 * best;
            sched[(long long)wg * max_items + cnt[best]] = tl[a].code;
            ++cnt[best];
            load[best] += tl[a].nk + 2;
        }
    }
    return 0;
}

extern "C" int jamie_gemm_bf16_ring(const jamie_gemm_problem* pr, int count, const int32_t* sched /*device*/, int n_wg, int max_items,
                                    const float* rg, void* rg_bf16, const long long* offsets, const long long* lengths, int n_ranges,
                                    float* partials, int n_partials, uint64_t* state, const jamie_latent_m* fin,
                                    unsigned* err /*device*/, void* stream) {
    JAMIE_ARG(pr != nullptr && sched != nullptr && err != nullptr && count >= 1 && count <= JAMIE_MAX_GEMM_GROUP, "1 <= count <= JAMIE_MAX_GEMM_GROUP");
    JAMIE_ARG(n_wg >= 8 && n_wg % 8 == 0 && max_items >= 1 && max_items <= RG_MAX_ITEMS, "n_wg a multiple of 8, max_items <= 48");
    RingGroup g;
    int rc = ring_fill(pr, count, &g, nullptr);
    if (rc) return rc;
    RangeRide rr;
    memset(&rr, 0, sizeof(rr));
    int blocks = 0;
    if (rg != nullptr) {
        rc = jamie_range_ride_fill(rg, rg_bf16, offsets, lengths, n_ranges, partials, n_partials, state, fin, &rr, &blocks);
        if (rc) return rc;
        JAMIE_ARG(blocks <= n_wg, "more range chunks than workgroups");
    }
    hipLaunchKernelGGL((gemm_bf16_ring_kernel<RG_NB, RG_LA>), dim3(n_wg), dim3((RG_NLOAD + 8) * 64), 0, (hipStream_t)stream, g, sched, max_items, rr, blocks, err);
    return jamie_launch_status("jamie_gemm_bf16_ring");
}

#endif  // JAMIE_EXPERIMENTS

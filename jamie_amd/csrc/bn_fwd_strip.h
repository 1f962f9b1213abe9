// BatchNorm1d (training statistics) + LeakyReLU + Dropout forward of ONE strip of 16 feature columns x all batch rows, by a
// TEAM of 512 threads (8 waves): the body of `bn_act_fwd4_kernel` (bn_act.hip), shared with the bf16 GEMM launch that reduces
// its own split-K slabs (gemm_bf16.hip, `jamie_gemm_bf16_bn`) so that both produce the same bits.
// Replaces native_batch_norm / leaky_relu / bernoulli_ + mul (reference model.py:152-154,162-164,193-195,198-200).
//
// A thread owns 4 consecutive columns x R rows (rows rp + 128 j): 16-byte loads / stores (1 KiB per wave-instruction = 16 rows
// x 64 B), up to three split-K slabs in flight per round trip, column sums by xor-shuffles over a wave's 16 row phases plus one
// 8-wave LDS step, one Philox call per pair of rows for the thread's 4 columns.
#pragma once
#include "common.h"

#define BN_CW 16
// the summed pre-activation (next read by the backward pass) stored non-temporally: -2 us per step (r03_ab_more_nt.log)
#define BN_OOB 0xFFFFFFF0u
#define BN4_RP 128
#define BN4_MAXR 4      // rows per thread of the default instance (B <= 512); the kernels are templates on R (4 or 8: B <= 1024)
#define BN4_NW 8        // waves per team (512 threads)
typedef unsigned int bn_u32x4 __attribute__((ext_vector_type(4)));

struct BnFwdDev {
    float* h; const float* gamma; const float* beta; float* rmean; float* rvar;
    float* smean; float* sinvstd; float* out; const uint8_t* mask;
    unsigned short* out_bf; unsigned short* outT_bf;
    long long slab_stride;
    int nslab, B, N, rng_stream, blk_begin;
    int panel;           // h (every slab, and the sum written back) in panels of P = JAMIE_PANEL columns: (row, col) at ((col / P) * B + row) * P + col % P
};

// Keep decision of element (row, col): 16 random bits against a 16-bit threshold.  One Philox call serves the 4
// consecutive columns of a quad (word e = col % 4) in the two rows r and r + 128 (low / high half of the word): counter =
// (col / 4, row % 128 + 128 * (row / 256)), half = (row / 128) % 2 -- the 8 elements a thread of the float4 kernels owns
// in rows rp + 256 q and rp + 256 q + 128.  Halves the Philox work (3.7 us per launch at one call per 4 elements).
__device__ __forceinline__ Philox4 drop_rand4(const uint64_t* rng, int stream, int col, int row) {
    const uint32_t rk = (uint32_t)(row & 127) | ((uint32_t)(row >> 8) << 7);
    return jamie_rand4(rng, (uint32_t)stream, ((uint64_t)(uint32_t)(col >> 2) << 32) | (uint64_t)rk);
}
__device__ __forceinline__ uint32_t drop_threshold16(float p) {
    const float t = p * 65536.f;
    return t <= 0.f ? 0u : (t >= 65535.f ? 65535u : (uint32_t)t);
}
__device__ __forceinline__ bool drop_keep(const uint64_t* rng, int stream, int col, int row, uint32_t thr16) {
    return ((drop_rand4(rng, stream, col, row).v[col & 3] >> (16 * ((row >> 7) & 1))) & 0xFFFFu) >= thr16;
}

// AUX: cache policy bits of the buffer load (0 = default; 16 = sc1: served by L2 / memory, never by this CU's L1 -- the loads of
// bytes other workgroups of the same launch have just written, MI355X_MICROARCH.md 'inter-workgroup visibility')
template <int AUX = 0>
__device__ __forceinline__ float4 buf_f32x4(__amdgpu_buffer_rsrc_t r, unsigned off) {
    const bn_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, AUX);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ unsigned buf_u32(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return (unsigned)__builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0);
}
// `t`: thread index inside the team (0 .. 128 CQ - 1); CQ = column quads per strip (4: 16 columns on 512 threads, 2: 8 columns on
// 256); the barriers are workgroup barriers (every team of the workgroup calls in step)
template <int CQ = 4>
__device__ __forceinline__ float4 col_reduce4(float4 v, float (*sh)[4 * CQ], int t) {
#pragma unroll
    for (int m = CQ; m < 64; m <<= 1) {
        v.x += __shfl_xor(v.x, m); v.y += __shfl_xor(v.y, m); v.z += __shfl_xor(v.z, m); v.w += __shfl_xor(v.w, m);
    }
    const int lane = t & 63, wid = t >> 6, cq = t & (CQ - 1);
    __syncthreads();
    if (lane < CQ) { sh[wid][4 * lane] = v.x; sh[wid][4 * lane + 1] = v.y; sh[wid][4 * lane + 2] = v.z; sh[wid][4 * lane + 3] = v.w; }
    __syncthreads();
    float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < 2 * CQ; ++w)
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] += sh[w][4 * cq + e];
    return make_float4(s[0], s[1], s[2], s[3]);
}
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
    return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)a) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)b) << 16);
}
// bf16 outputs of a strip held as val[j] = 4 columns of row rp + 128 j: row-major [B, N] (8-byte stores) and, optionally,
// transposed [N, B] through the LDS tile `tl` (16 x (128 R + 2) shorts)
template <int R, int CQ = 4>
__device__ __forceinline__ void strip_out_bf16x4(const float4 (&val)[R], unsigned short* out_bf, unsigned short* outT_bf,
                                                 unsigned short* tl, int B, int N, int col0, int t, bool cok) {
    const int cq = t & (CQ - 1), rp = t / CQ;
    const int col = col0 + 4 * cq;
    if (out_bf && cok) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int row = rp + j * BN4_RP;
            if (row < B)
                *reinterpret_cast<uint2*>(out_bf + (long long)row * N + col) =
                    make_uint2(pack_bf16x2(val[j].x, val[j].y), pack_bf16x2(val[j].z, val[j].w));
        }
    }
    if (CQ != 4 || !outT_bf) return;                 // (the transposed copy is laid out for 16-column strips)
    __syncthreads();
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int row = rp + j * BN4_RP;
        tl[(4 * cq) * (128 * R + 2) + row] = __builtin_bit_cast(unsigned short, (__bf16)val[j].x);
        tl[(4 * cq + 1) * (128 * R + 2) + row] = __builtin_bit_cast(unsigned short, (__bf16)val[j].y);
        tl[(4 * cq + 2) * (128 * R + 2) + row] = __builtin_bit_cast(unsigned short, (__bf16)val[j].z);
        tl[(4 * cq + 3) * (128 * R + 2) + row] = __builtin_bit_cast(unsigned short, (__bf16)val[j].w);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < R / 2; ++i) {
        const int q = t + 512 * i, cc = q / (16 * R), r8 = (q % (16 * R)) * 8;
        if (col0 + cc < N && r8 < B) {        // B is a multiple of 8 in bf16 mode
            const unsigned* sp = reinterpret_cast<const unsigned*>(tl + cc * (128 * R + 2) + r8);
            *reinterpret_cast<uint4*>(outT_bf + (long long)(col0 + cc) * B + r8) = make_uint4(sp[0], sp[1], sp[2], sp[3]);
        }
    }
}

// Diagnostic build only (-DJAMIE_BN_STAMP, tools/stamp_bn.py): thread 0 of every workgroup stamps s_memrealtime at the phases of
// the forward strip (entry / slabs summed / statistics done / stores issued / stores retired); no stamp exists in the product build.
#ifdef JAMIE_BN_STAMP
#define BN_NSTAMP 8
extern __device__ unsigned long long jamie_bn_dbg_stamps[4096 * BN_NSTAMP];
#define BN_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 4096) jamie_bn_dbg_stamps[blockIdx.x * BN_NSTAMP + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define BN_STAMPV(k, v) do { if (threadIdx.x == 0 && blockIdx.x < 4096) jamie_bn_dbg_stamps[blockIdx.x * BN_NSTAMP + (k)] = (unsigned long long)(v); } while (0)
#else
#define BN_STAMP(k) do {} while (0)
#define BN_STAMPV(k, v) do {} while (0)
#endif

// One strip: columns col0 .. col0 + 15 (those < N; `active` false: the team only takes part in the barriers).
// The summed pre-activation goes back to slab 0 (read again by the backward pass), the batch statistics to save_mean /
// save_invstd, the running statistics are updated in place, the activation goes out as fp32 and / or bf16.
template <int R, int AUX, int CQ = 4>
__device__ __forceinline__ void bn_fwd4_strip(const BnFwdDev& P, int col0, int t, bool active, float (*sh)[4 * CQ],
                                              unsigned short* tl, float p_drop, float momentum, float eps, float slope,
                                              const uint64_t* rng) {
    const int cq = t & (CQ - 1), rp = t / CQ;
    const int col = col0 + 4 * cq;
    const bool cok = active && col < P.N;             // N % 4 == 0: a quad is wholly in or out
    BN_STAMP(0);
    BN_STAMPV(5, __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) * 1000 + __builtin_amdgcn_s_getreg(((8 - 1) << 11) | (8 << 6) | 4));
    BN_STAMPV(6, P.N * 10 + P.nslab);
    const int B = P.B, N = P.N, nslab = P.nslab;
    // panel layout (round 5; jamie_hip.h: JAMIE_PANEL): the columns of a panel are contiguous for every row, the panels of a
    // slab follow one another -- this strip's rows are whole blocks of B x 4 JAMIE_PANEL bytes per slab instead of
    // B segments 4 N bytes apart: the slab loads, the statistics' only input, arrive 1.5-2 us earlier per launch
    // (profiles/r05_ab_bn_panel_timing.log)
    const bool pan = P.panel != 0;
    constexpr unsigned PW = JAMIE_PANEL, PB = 4u * JAMIE_PANEL;
    const unsigned row_bytes = pan ? PB : (unsigned)N * 4u, slab_bytes = (unsigned)(P.slab_stride * 4);
    const unsigned one_slab = pan ? (unsigned)((N + PW - 1) / PW) * (unsigned)B * PB : (unsigned)B * (unsigned)N * 4u;
    const __amdgpu_buffer_rsrc_t h_rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)P.h, 0, (int)((unsigned)(nslab - 1) * slab_bytes + one_slab), 0x00020000);
    const __amdgpu_buffer_rsrc_t m_rs = __builtin_amdgcn_make_buffer_rsrc((void*)P.mask, 0, P.mask ? B * N : 0, 0x00020000);
    const unsigned coff = pan ? ((unsigned)col / PW) * ((unsigned)B * PB) + ((unsigned)col % PW) * 4u : (unsigned)col * 4u;
    unsigned roff[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int row = rp + j * BN4_RP;
        roff[j] = (row < B && cok) ? (unsigned)row * row_bytes + coff : BN_OOB;
    }
    // latency order: parameter loads and the first slabs are issued first; the Philox keep words (pure VALU,
    // ~100 instructions per call) are computed while those loads are in flight (they cost 3.7 us per launch when they
    // sat behind the statistics: rocprofv3, tools/trace_bn.sh)
    float ga[4] = {0.f, 0.f, 0.f, 0.f}, be[4] = {0.f, 0.f, 0.f, 0.f}, rm_old[4] = {0.f, 0.f, 0.f, 0.f}, rv_old[4] = {0.f, 0.f, 0.f, 0.f};
    if (cok) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { ga[e] = P.gamma[col + e]; be[e] = P.beta[col + e]; }
        if (rp == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { rm_old[e] = P.rmean[col + e]; rv_old[e] = P.rvar[col + e]; }
        }
    }
    const bool drop = p_drop > 0.f;
    const float keep_scale = drop ? 1.f / (1.f - p_drop) : 1.f;
    const uint32_t thr = drop_threshold16(p_drop);
    unsigned mk[R];
    if (drop && P.mask) {
#pragma unroll
        for (int j = 0; j < R; ++j)       // (the explicit masks of the parity tests are row-major [B, N] bytes)
            mk[j] = buf_u32(m_rs, roff[j] == BN_OOB ? BN_OOB : (unsigned)(rp + j * BN4_RP) * (unsigned)N + (unsigned)col);
    }
    float4 v[R];
#pragma unroll
    for (int j = 0; j < R; ++j) v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned keepbits = 0xFFFFFFFFu;              // bit 4 j + e: element (row j, column e) of this thread is kept
    // (R == 4: THREE slabs per round trip -- the forward launches of config 2 have (3, 2) K slices, and a third slab in a second
    //  trip was one more memory latency for every workgroup of the larger modality; R == 8 keeps two: registers)
    constexpr int TRIP = R <= 4 ? 3 : 2;
    for (int s = 0; s < nslab; s += TRIP) {
        float4 a[R], b[R], c[TRIP > 2 ? R : 1];
        const bool two = s + 1 < nslab, three = TRIP > 2 && s + 2 < nslab;
#pragma unroll
        for (int j = 0; j < R; ++j) a[j] = buf_f32x4<AUX>(h_rs, roff[j] == BN_OOB ? BN_OOB : roff[j] + (unsigned)s * slab_bytes);
#pragma unroll
        for (int j = 0; j < R; ++j) b[j] = buf_f32x4<AUX>(h_rs, (roff[j] == BN_OOB || !two) ? BN_OOB : roff[j] + (unsigned)(s + 1) * slab_bytes);
        if constexpr (TRIP > 2) {
#pragma unroll
            for (int j = 0; j < R; ++j) c[j] = buf_f32x4<AUX>(h_rs, (roff[j] == BN_OOB || !three) ? BN_OOB : roff[j] + (unsigned)(s + 2) * slab_bytes);
        }
        if (s == 0 && drop) {                     // VALU work under the loads just issued
            keepbits = 0u;
#pragma unroll
            for (int j = 0; j < R; ++j) {
                if (P.mask) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) keepbits |= (((mk[j] >> (8 * e)) & 0xFFu) != 0 ? 1u : 0u) << (4 * j + e);
                } else if ((j & 1) == 0) {        // rows rp + 128 j and rp + 128 (j + 1): low / high halves of one call
                    const Philox4 r = drop_rand4(rng, P.rng_stream, col, rp + j * BN4_RP);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        keepbits |= ((r.v[e] & 0xFFFFu) >= thr ? 1u : 0u) << (4 * j + e);
                        keepbits |= ((r.v[e] >> 16) >= thr ? 1u : 0u) << (4 * (j + 1) + e);
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < R; ++j) {
            v[j].x += a[j].x; v[j].y += a[j].y; v[j].z += a[j].z; v[j].w += a[j].w;
            v[j].x += b[j].x; v[j].y += b[j].y; v[j].z += b[j].z; v[j].w += b[j].w;
            if constexpr (TRIP > 2) { v[j].x += c[j].x; v[j].y += c[j].y; v[j].z += c[j].z; v[j].w += c[j].w; }
        }
    }
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < R; ++j) { sum.x += v[j].x; sum.y += v[j].y; sum.z += v[j].z; sum.w += v[j].w; }
#ifdef JAMIE_BN_STAMP
    if (sum.x == 1.2345e-30f) BN_STAMP(7);      // (keeps the sums live in front of the stamp: the loads have landed)
    BN_STAMP(1);
#endif
    float4 mean = col_reduce4<CQ>(sum, sh, t);
    const float fB = (float)B;
    mean.x /= fB; mean.y /= fB; mean.z /= fB; mean.w /= fB;
    float4 sq = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < R; ++j) {
        if (rp + j * BN4_RP < B) {
            const float dx = v[j].x - mean.x, dy = v[j].y - mean.y, dz = v[j].z - mean.z, dw = v[j].w - mean.w;
            sq.x += dx * dx; sq.y += dy * dy; sq.z += dz * dz; sq.w += dw * dw;
        }
    }
    float4 var = col_reduce4<CQ>(sq, sh, t);
    var.x /= fB; var.y /= fB; var.z /= fB; var.w /= fB;
    const float4 invstd = make_float4(rsqrtf(var.x + eps), rsqrtf(var.y + eps), rsqrtf(var.z + eps), rsqrtf(var.w + eps));
    BN_STAMP(2);
    if (cok && rp == 0) {
        const float mv[4] = {mean.x, mean.y, mean.z, mean.w}, vv[4] = {var.x, var.y, var.z, var.w};
        const float iv[4] = {invstd.x, invstd.y, invstd.z, invstd.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            P.smean[col + e] = mv[e];
            P.sinvstd[col + e] = iv[e];
            const float unb = B > 1 ? vv[e] * ((float)B / (float)(B - 1)) : vv[e];
            P.rmean[col + e] = (1.f - momentum) * rm_old[e] + momentum * mv[e];
            P.rvar[col + e] = (1.f - momentum) * rv_old[e] + momentum * unb;
        }
    }
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int row = rp + j * BN4_RP;
        float y[4] = {0.f, 0.f, 0.f, 0.f};
        if (row < B && cok) {
            const long long o = (long long)row * N + col;
            if (nslab > 1) {      // (the summed pre-activation: next read by the backward pass, stored non-temporally)
                // ONE 16-byte non-temporal store (a float4 vector: four scalar stores only merge while hipcc can prove the
                // pointer's alignment -- behind the layout select it could not, and the store phase of the launch went from
                // 1.3 to 3.0 us: profiles/r05_stamps_bn_fwd_panel.log)
                typedef float bn_f32x4 __attribute__((ext_vector_type(4)));
                bn_f32x4* hp = reinterpret_cast<bn_f32x4*>(P.h + (pan ? (long long)(roff[j] >> 2) : o));
                __builtin_nontemporal_store((bn_f32x4){v[j].x, v[j].y, v[j].z, v[j].w}, hp);
            }
            const float hv[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
            const float mv[4] = {mean.x, mean.y, mean.z, mean.w}, iv[4] = {invstd.x, invstd.y, invstd.z, invstd.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float u = (hv[e] - mv[e]) * iv[e] * ga[e] + be[e];
                u = u > 0.f ? u : slope * u;
                if (drop) u = ((keepbits >> (4 * j + e)) & 1u) ? u * keep_scale : 0.f;
                y[e] = u;
            }
            if (P.out) *reinterpret_cast<float4*>(P.out + o) = make_float4(y[0], y[1], y[2], y[3]);
        }
        v[j] = make_float4(y[0], y[1], y[2], y[3]);
    }
    if (P.out_bf || P.outT_bf) strip_out_bf16x4<R, CQ>(v, P.out_bf, P.outT_bf, tl, B, N, col0, t, cok);
#ifdef JAMIE_BN_STAMP
    BN_STAMP(3);
    __builtin_amdgcn_s_waitcnt(0);           // (vmcnt = lgkmcnt = expcnt = 0: the stores have been acknowledged)
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
    BN_STAMP(4);
#endif
}

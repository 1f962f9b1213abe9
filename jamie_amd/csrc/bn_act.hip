// BatchNorm1d (training statistics) + LeakyReLU + Dropout, forward and backward, gfx950.
//
// Replaces native_batch_norm / leaky_relu / bernoulli_+mul and their autograd backward
// (reference model.py:152-154,162-164,193-195,198-200; jamie.py:734).
//
// One workgroup owns a strip of CW = 16 feature columns for ALL batch rows, so the batch statistics
// (exact two-pass mean / biased variance, like ATen's CPU kernel) are local to the workgroup:
// 256 threads = 16 columns x 16 row phases; with B <= 512 every thread keeps its <= 32 values in
// registers and h is read from HBM/L2 exactly once.  Larger batches re-read h.
// All loads are raw buffer loads: a row/column beyond the matrix is an out-of-range offset that returns 0,
// so the 32 loads of a thread are issued back to back with no branches and one wait.
// The pre-BN activations may arrive as split-K slabs; they are summed on the fly and the sum is written
// back to slab 0 for the backward pass.  Dropout masks come from Philox4x32-10 keyed by (seed, step,
// stream) with one call per 4 elements, and are regenerated identically in the backward kernel; tests
// pass explicit masks.
#include "common.h"
#include "colsum.h"

#include "bn_fwd_strip.h"

#ifdef JAMIE_BN_STAMP
__device__ unsigned long long jamie_bn_dbg_stamps[4096 * BN_NSTAMP];
extern "C" int jamie_debug_bn_stamps(unsigned long long* host_out, int n_blocks) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(jamie_bn_dbg_stamps), sizeof(unsigned long long) * BN_NSTAMP * n_blocks);
}
#endif

// cache policy bits of the float4 kernels' once-read loads (split-K slabs, saved pre-activations, upstream gradients): 2 = nt
// (they are dead once read; default policy, 0: the step 602.6 instead of 599.4 us on one box, three interleaved rounds,
// profiles/r03_ab_bn_nt_loads.log)
#define JAMIE_BN_LD_AUX 2

// Strip order: a problem gets 8 * ceil(strips / 8) workgroups and workgroup lb handles strip (lb & 7) * q + (lb >> 3),
// q = ceil(strips / 8): blocks are dealt round-robin over the 8 XCDs (lb & 7), so every XCD owns a CONTIGUOUS range of
// strips.  Neighbouring strips share 128-byte lines (a strip is 64 bytes of an fp32 row, 32 bytes of a bf16 row): the
// two halves of a line are then read through ONE L2 and the partial-line stores of neighbours merge there.
__device__ __forceinline__ int bn_strip(int lb, int n_cols, int width = 16) {
    const int nst = (n_cols + width - 1) / width, q = (nst + 7) >> 3;
    return (lb & 7) * q + (lb >> 3);
}
#define BN_RP 16
#define BN_MAXR 32

struct BnFwdGroup { BnFwdDev p[JAMIE_MAX_GROUP]; int count; };

struct BnBwdDev {
    float* da; const float* h; const float* gamma; const float* beta; const float* smean;
    const float* sinvstd; float* dgamma; float* dbeta; float* dbias; const uint8_t* mask;
    unsigned short* dh_bf; unsigned short* dhT_bf;
    long long slab_stride;
    int nslab, B, N, rng_stream, accumulate, blk_begin, skip_f32;
    int panel;           // da (every slab) and h in panels of 16 columns (bn_fwd_strip.h); dh leaves as bf16 row-major only
};
struct BnBwdGroup { BnBwdDev p[JAMIE_MAX_GROUP]; int count; };

// sum over the 16 row phases of one column; every thread of the column gets the total
__device__ __forceinline__ float col_reduce(float v, float (*sh)[BN_CW + 1], int rp, int c) {
    __syncthreads();
    sh[rp][c] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < BN_RP; ++i) t += sh[i][c];
    return t;
}

__device__ __forceinline__ float buf_f32(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0));
}
__device__ __forceinline__ unsigned buf_u8(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return (unsigned)__builtin_amdgcn_raw_buffer_load_b8(r, (int)off, 0, 0);
}

// bf16 outputs of a cached strip (thread (c, rp) holds rows rp + 16 j of column c in val[j]):
//   row-major  [B, N]: 2-byte stores (16 columns = 32-byte segments per row);
//   transposed [N, B]: the strip is 16 whole rows of the transposed matrix; staged through LDS (row stride 514
//   elements: conflict-free 2-byte writes) and written as 16-byte stores, 1 KiB contiguous per column.
#define BN_TS 514
__device__ __forceinline__ void strip_out_bf16(const float (&val)[BN_MAXR], unsigned short* out_bf, unsigned short* outT_bf,
                                               unsigned short* tl, int B, int N, int col0, int c, int rp, bool cok) {
    const int col = col0 + c;
    if (out_bf && cok) {
#pragma unroll
        for (int j = 0; j < BN_MAXR; ++j) {
            const int row = rp + j * BN_RP;
            if (row < B) out_bf[(long long)row * N + col] = __builtin_bit_cast(unsigned short, (__bf16)val[j]);
        }
    }
    if (!outT_bf) return;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < BN_MAXR; ++j) tl[c * BN_TS + rp + j * BN_RP] = __builtin_bit_cast(unsigned short, (__bf16)val[j]);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = threadIdx.x + 256 * i, cc = q >> 6, r8 = (q & 63) * 8;
        if (col0 + cc < N && r8 < B) {        // B is a multiple of 8 in bf16 mode
            const unsigned* sp = reinterpret_cast<const unsigned*>(tl + cc * BN_TS + r8);
            *reinterpret_cast<uint4*>(outT_bf + (long long)(col0 + cc) * B + r8) = make_uint4(sp[0], sp[1], sp[2], sp[3]);
        }
    }
}

template <bool CACHED>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(BnFwdGroup g, float p_drop, float momentum, float eps,
                                                         float slope, const uint64_t* rng) {
    __shared__ float sh[BN_RP][BN_CW + 1];
    __shared__ __attribute__((aligned(16))) unsigned short tl[CACHED ? BN_CW * BN_TS : 8];
    int pi = 0;
#pragma unroll
    for (int i = 1; i < JAMIE_MAX_GROUP; ++i)
        if (i < g.count && (int)blockIdx.x >= g.p[i].blk_begin) pi = i;
    const BnFwdDev& P = g.p[pi];
    const int c = threadIdx.x % BN_CW, rp = threadIdx.x / BN_CW;
    const int col = bn_strip((int)blockIdx.x - P.blk_begin, P.N) * BN_CW + c;
    const bool cok = col < P.N;
    const int B = P.B, N = P.N;
    const unsigned row_bytes = (unsigned)N * 4u, slab_bytes = (unsigned)(P.slab_stride * 4);
    const __amdgpu_buffer_rsrc_t h_rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)P.h, 0, (int)((unsigned)(P.nslab - 1) * slab_bytes + (unsigned)B * row_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t m_rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)P.mask, 0, P.mask ? B * N : 0, 0x00020000);
    const unsigned col_off = cok ? (unsigned)col * 4u : BN_OOB;
    const int nslab = P.nslab;

    float v[CACHED ? BN_MAXR : 1];
    float sum = 0.f;
    if (CACHED) {
#pragma unroll
        for (int j = 0; j < BN_MAXR; ++j) v[j] = 0.f;
        for (int s = 0; s < nslab; ++s) {
#pragma unroll
            for (int j = 0; j < BN_MAXR; ++j) {
                const int row = rp + j * BN_RP;
                const unsigned off = (row < B && cok) ? (unsigned)s * slab_bytes + (unsigned)row * row_bytes + col_off : BN_OOB;
                v[j] += buf_f32(h_rs, off);
            }
        }
#pragma unroll
        for (int j = 0; j < BN_MAXR; ++j) sum += v[j];
    } else {
        for (int row = rp; row < B; row += BN_RP)
            for (int s = 0; s < nslab; ++s)
                sum += buf_f32(h_rs, cok ? (unsigned)s * slab_bytes + (unsigned)row * row_bytes + col_off : BN_OOB);
    }
    const float mean = col_reduce(sum, sh, rp, c) / (float)B;
    float sq = 0.f;
    if (CACHED) {
#pragma unroll
        for (int j = 0; j < BN_MAXR; ++j) {
            const float d = v[j] - mean;
            if (rp + j * BN_RP < B) sq += d * d;
        }
    } else {
        for (int row = rp; row < B; row += BN_RP) {
            float hv = 0.f;
            for (int s = 0; s < nslab; ++s)
                hv += buf_f32(h_rs, cok ? (unsigned)s * slab_bytes + (unsigned)row * row_bytes + col_off : BN_OOB);
            sq += (hv - mean) * (hv - mean);
        }
    }
    const float var = col_reduce(sq, sh, rp, c) / (float)B;   // biased
    const float invstd = rsqrtf(var + eps);
    if (!CACHED && !cok) return;
    if (cok && rp == 0) {
        P.smean[col] = mean;
        P.sinvstd[col] = invstd;
        const float unb = B > 1 ? var * ((float)B / (float)(B - 1)) : var;
        P.rmean[col] = (1.f - momentum) * P.rmean[col] + momentum * mean;
        P.rvar[col] = (1.f - momentum) * P.rvar[col] + momentum * unb;
    }
    const float ga = cok ? P.gamma[col] : 0.f, be = cok ? P.beta[col] : 0.f;
    const bool drop = p_drop > 0.f;
    const float keep_scale = drop ? 1.f / (1.f - p_drop) : 1.f;
    const uint32_t thr = drop_threshold16(p_drop);
    if (CACHED) {
        unsigned mk[BN_MAXR];
        if (drop && P.mask) {
#pragma unroll
            for (int j = 0; j < BN_MAXR; ++j) {
                const int row = rp + j * BN_RP;
                mk[j] = buf_u8(m_rs, (row < B && cok) ? (unsigned)row * (unsigned)N + (unsigned)col : BN_OOB);
            }
        }
#pragma unroll
        for (int j = 0; j < BN_MAXR; ++j) {
            const int row = rp + j * BN_RP;
            float y = 0.f;
            if (row < B && cok) {
                const long long o = (long long)row * N + col;
                if (nslab > 1) P.h[o] = v[j];
                y = (v[j] - mean) * invstd * ga + be;
                y = y > 0.f ? y : slope * y;
                if (drop) {
                    const bool keep = P.mask ? (mk[j] != 0) : drop_keep(rng, P.rng_stream, col, row, thr);
                    y = keep ? y * keep_scale : 0.f;
                }
                if (P.out) P.out[o] = y;
            }
            v[j] = y;
        }
        if constexpr (CACHED) {
            if (P.out_bf || P.outT_bf)
                strip_out_bf16(v, P.out_bf, P.outT_bf, tl, B, N, bn_strip((int)blockIdx.x - P.blk_begin, P.N) * BN_CW, c, rp, cok);
        }
    } else {
        for (int row = rp; row < B; row += BN_RP) {
            float hv = 0.f;
            for (int s = 0; s < nslab; ++s)
                hv += buf_f32(h_rs, (unsigned)s * slab_bytes + (unsigned)row * row_bytes + col_off);
            const long long o = (long long)row * N + col;
            if (nslab > 1) P.h[o] = hv;
            float y = (hv - mean) * invstd * ga + be;
            y = y > 0.f ? y : slope * y;
            if (drop) {
                bool keep;
                if (P.mask) keep = P.mask[o] != 0;
                else keep = drop_keep(rng, P.rng_stream, col, row, thr);
                y = keep ? y * keep_scale : 0.f;
            }
            if (P.out) P.out[o] = y;
        }
    }
}

template <bool CACHED>
__global__ __launch_bounds__(256) void bn_act_bwd_kernel(BnBwdGroup g, float p_drop, float slope,
                                                         const uint64_t* rng) {
    __shared__ float sh[BN_RP][BN_CW + 1];
    __shared__ __attribute__((aligned(16))) unsigned short tl[CACHED ? BN_CW * BN_TS : 8];
    int pi = 0;
#pragma unroll
    for (int i = 1; i < JAMIE_MAX_GROUP; ++i)
        if (i < g.count && (int)blockIdx.x >= g.p[i].blk_begin) pi = i;
    const BnBwdDev& P = g.p[pi];
    const int c = threadIdx.x % BN_CW, rp = threadIdx.x / BN_CW;
    const int col = bn_strip((int)blockIdx.x - P.blk_begin, P.N) * BN_CW + c;
    const bool cok = col < P.N;
    const int B = P.B, N = P.N;
    const float mean = cok ? P.smean[col] : 0.f, invstd = cok ? P.sinvstd[col] : 0.f;
    const float ga = cok ? P.gamma[col] : 0.f, be = cok ? P.beta[col] : 0.f;
    const bool drop = p_drop > 0.f;
    const float keep_scale = drop ? 1.f / (1.f - p_drop) : 1.f;
    const uint32_t thr = drop_threshold16(p_drop);
    const unsigned row_bytes = (unsigned)N * 4u, slab_bytes = (unsigned)(P.slab_stride * 4);
    const __amdgpu_buffer_rsrc_t d_rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)P.da, 0, (int)((unsigned)(P.nslab - 1) * slab_bytes + (unsigned)B * row_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t h_rs = __builtin_amdgcn_make_buffer_rsrc((void*)P.h, 0, (int)((unsigned)B * row_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t m_rs = __builtin_amdgcn_make_buffer_rsrc((void*)P.mask, 0, P.mask ? B * N : 0, 0x00020000);
    const unsigned col_off = cok ? (unsigned)col * 4u : BN_OOB;
    const int nslab = P.nslab;

    // dy (grad wrt BN output) from the activation gradient d, the normalised input xn and the keep decision
    auto to_dy = [&](float d, float xn, bool keep) -> float {
        const float y = xn * ga + be;
        if (drop) d = keep ? d * keep_scale : 0.f;
        return y > 0.f ? d : d * slope;
    };
    // non-cached element: everything re-read
    auto elem_slow = [&](int row, float& dy, float& xn) {
        const long long o = (long long)row * N + col;
        float d = 0.f;
        for (int s = 0; s < nslab; ++s) d += P.da[o + s * P.slab_stride];
        xn = (P.h[o] - mean) * invstd;
        bool keep = true;
        if (drop) {
            keep = P.mask ? (P.mask[o] != 0) : drop_keep(rng, P.rng_stream, col, row, thr);
        }
        dy = to_dy(d, xn, keep);
    };

    float dyv[CACHED ? BN_MAXR : 1], xnv[CACHED ? BN_MAXR : 1];
    float s1 = 0.f, s2 = 0.f;
    if (CACHED) {
#pragma unroll
        for (int j = 0; j < BN_MAXR; ++j) dyv[j] = 0.f;
        for (int s = 0; s < nslab; ++s) {
#pragma unroll
            for (int j = 0; j < BN_MAXR; ++j) {
                const int row = rp + j * BN_RP;
                dyv[j] += buf_f32(d_rs, (row < B && cok) ? (unsigned)s * slab_bytes + (unsigned)row * row_bytes + col_off : BN_OOB);
            }
        }
#pragma unroll
        for (int j = 0; j < BN_MAXR; ++j) {
            const int row = rp + j * BN_RP;
            xnv[j] = buf_f32(h_rs, (row < B && cok) ? (unsigned)row * row_bytes + col_off : BN_OOB);
        }
        unsigned mk[BN_MAXR];
        if (drop && P.mask) {
#pragma unroll
            for (int j = 0; j < BN_MAXR; ++j) {
                const int row = rp + j * BN_RP;
                mk[j] = buf_u8(m_rs, (row < B && cok) ? (unsigned)row * (unsigned)N + (unsigned)col : BN_OOB);
            }
        }
#pragma unroll
        for (int j = 0; j < BN_MAXR; ++j) {
            const int row = rp + j * BN_RP;
            const bool keep = drop ? (P.mask ? (mk[j] != 0) : drop_keep(rng, P.rng_stream, col, row, thr)) : true;
            const bool ok = cok && row < B;
            const float xn = ok ? (xnv[j] - mean) * invstd : 0.f;
            xnv[j] = xn;
            dyv[j] = ok ? to_dy(dyv[j], xn, keep) : 0.f;
            s1 += dyv[j];
            s2 += dyv[j] * xn;
        }
    } else {
        if (cok)
            for (int row = rp; row < B; row += BN_RP) {
                float dy, xn;
                elem_slow(row, dy, xn);
                s1 += dy;
                s2 += dy * xn;
            }
    }
    const float dbeta = col_reduce(s1, sh, rp, c);
    const float dgamma = col_reduce(s2, sh, rp, c);
    const float invB = 1.f / (float)B;
    const float k1 = dbeta * invB, k2 = dgamma * invB, gi = ga * invstd;
    float s3 = 0.f;
    if (CACHED) {
#pragma unroll
        for (int j = 0; j < BN_MAXR; ++j) {
            const int row = rp + j * BN_RP;
            float dh = 0.f;
            if (cok && row < B) {
                dh = gi * (dyv[j] - k1 - xnv[j] * k2);
                if (!P.skip_f32) P.da[(long long)row * N + col] = dh;
                s3 += dh;
            }
            dyv[j] = dh;
        }
        if constexpr (CACHED) {
            if (P.dh_bf || P.dhT_bf)
                strip_out_bf16(dyv, P.dh_bf, P.dhT_bf, tl, B, N, bn_strip((int)blockIdx.x - P.blk_begin, P.N) * BN_CW, c, rp, cok);
        }
    } else {
        if (cok)
            for (int row = rp; row < B; row += BN_RP) {
                float dy, xn;
                elem_slow(row, dy, xn);
                const float dh = gi * (dy - k1 - xn * k2);
                P.da[(long long)row * N + col] = dh;   // slab 0 <- dh (slab 0 is only read by this thread)
                s3 += dh;
            }
    }
    const float dbias = col_reduce(s3, sh, rp, c);
    if (cok && rp == 0) {
        if (P.accumulate) {
            P.dgamma[col] += dgamma;
            P.dbeta[col] += dbeta;
            if (P.dbias) P.dbias[col] += dbias;
        } else {
            P.dgamma[col] = dgamma;
            P.dbeta[col] = dbeta;
            if (P.dbias) P.dbias[col] = dbias;
        }
    }
}


// ------------------------------------------------------------------------------------------------
// float4 variants (N % 4 == 0, B <= 512: every training shape of the bf16 / fp32 step).  The dword-per-lane kernels
// above move 256 B per wave-instruction and ran at 1.7-2 TB/s (rocprofv3, config 2: 23 us for ~40 MB): the
// address path handles a wave-instruction in >= 16 cycles whatever its width.  Here a thread owns 4 consecutive
// columns x 4 rows (rows rp + 128 j, rp = 0..127; 512 threads): 16-byte loads / stores (1 KiB per wave-instruction = 16 rows x 64 B),
// two slabs in flight at a time, column sums by xor-shuffles over the 16 row phases of a wave + a 4-wave LDS step.
// ------------------------------------------------------------------------------------------------
// two column sums at once (one pair of barriers): sh2 is [4][2 * BN_CW]
template <int CQ = 4>
__device__ __forceinline__ void col_reduce4x2(float4& a, float4& b, float (*sh2)[8 * CQ], int tid) {
#pragma unroll
    for (int m = CQ; m < 64; m <<= 1) {
        a.x += __shfl_xor(a.x, m); a.y += __shfl_xor(a.y, m); a.z += __shfl_xor(a.z, m); a.w += __shfl_xor(a.w, m);
        b.x += __shfl_xor(b.x, m); b.y += __shfl_xor(b.y, m); b.z += __shfl_xor(b.z, m); b.w += __shfl_xor(b.w, m);
    }
    const int lane = tid & 63, wid = tid >> 6, cq = tid & (CQ - 1);
    __syncthreads();
    if (lane < CQ) {
        float* d = sh2[wid] + 8 * lane;
        d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w; d[4] = b.x; d[5] = b.y; d[6] = b.z; d[7] = b.w;
    }
    __syncthreads();
    float t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        t[e] = 0.f;
#pragma unroll
        for (int w = 0; w < 2 * CQ; ++w) t[e] += sh2[w][8 * cq + e];
    }
    a = make_float4(t[0], t[1], t[2], t[3]);
    b = make_float4(t[4], t[5], t[6], t[7]);
}
// PREFETCH RIDER (jamie_bn_act_fwd_pf / jamie_bn_act_bwd_pf): extra workgroups of a BatchNorm launch -- a latency-bound
// launch with memory bandwidth to spare -- read the weights the NEXT launch's product streams (default cache policy: the lines
// land in the Infinity Cache), so that GEMM starts on warm weights instead of HBM-cold ones.  Loads only; nothing is written.
#define BN_PF_BLOCKS 64
#ifndef BN_PF_UNROLL
#define BN_PF_UNROLL 4
#endif
static int bn_pf_blocks() { return BN_PF_BLOCKS; }     // (swept in round 3: 16 / 32 stretch the launch, 128 / 256 no better)
#define BN_PF_MAX 8
struct PfRanges { const char* p[BN_PF_MAX]; long long bytes[BN_PF_MAX]; int n; };
__device__ __forceinline__ void prefetch_range(const char* p, long long bytes, int blk, int nblk) {
    const long long stride = (long long)nblk * blockDim.x * 16;
    for (long long off = ((long long)blk * blockDim.x + threadIdx.x) * 16; off < bytes; off += BN_PF_UNROLL * stride) {
        bn_u32x4 v[BN_PF_UNROLL];
#pragma unroll
        for (int u = 0; u < BN_PF_UNROLL; ++u) {
            const long long o = off + u * stride;
            v[u] = o + 16 <= bytes ? *reinterpret_cast<const bn_u32x4*>(p + o) : bn_u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int u = 0; u < BN_PF_UNROLL; ++u) asm volatile("" ::"v"(v[u]));
    }
}

__device__ __forceinline__ void prefetch_block(const PfRanges& r, int blk, int nblk) {
#pragma unroll
    for (int i = 0; i < BN_PF_MAX; ++i)
        if (i < r.n) prefetch_range(r.p[i], r.bytes[i], blk, nblk);
}

// (A/B knob -DJAMIE_BN_FWD_WAVES=4: at most 128 VGPRs for R = 4, i.e. two 512-thread workgroups per CU instead of one -- the
//  compiler's own allocation is 130 registers; measured slower, also with the second-slot workgroups holding their loads back
//  by 3-6 us: DESIGN.md §4, profiles/r03_stamps_bn_fwd*.log, r03_ab_bn_stagger_rejected.log)
#define JAMIE_BN_FWD_WAVES 1
template <int R, int CQ>
__global__ __launch_bounds__(128 * CQ, (R <= 4 ? JAMIE_BN_FWD_WAVES : 1)) void bn_act_fwd4_kernel(BnFwdGroup g, float p_drop, float momentum, float eps,
                                                               float slope, const uint64_t* rng, PfRanges pf, int n_main) {
    if ((int)blockIdx.x >= n_main) {
        prefetch_block(pf, (int)blockIdx.x - n_main, (int)gridDim.x - n_main);
        return;
    }
    __shared__ float shraw[2 * CQ * 4 * CQ];                     // [waves = 2 CQ][columns = 4 CQ]
    float (*sh)[4 * CQ] = reinterpret_cast<float (*)[4 * CQ]>(shraw);
    __shared__ __attribute__((aligned(16))) unsigned short tl[CQ == 4 ? BN_CW * (128 * R + 2) : 8];
    // (every problem's first workgroup loaded up front and the chosen problem's descriptor by value: as guarded iterations and
    //  fields fetched at their first use these were eight dependent scalar-memory round trips in front of the first slab load of a
    //  launch that lasts 11-17 us)
    int bb[JAMIE_MAX_GROUP];
#pragma unroll
    for (int i = 0; i < JAMIE_MAX_GROUP; ++i) bb[i] = g.p[i].blk_begin;
    const int cnt = g.count;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < JAMIE_MAX_GROUP; ++i) pi = (i < cnt && (int)blockIdx.x >= bb[i]) ? i : pi;
    const BnFwdDev P = g.p[pi];
    const int col0 = bn_strip((int)blockIdx.x - P.blk_begin, P.N, 4 * CQ) * (4 * CQ);
    bn_fwd4_strip<R, JAMIE_BN_LD_AUX, CQ>(P, col0, (int)threadIdx.x, true, sh, tl, p_drop, momentum, eps, slope, rng);      // (bn_fwd_strip.h)
}

// `cs` / `cs_begin`: workgroups cs_begin .. are EXTRA ones that compute column sums (jamie_bn_act_bwd_cs: the decoder's
// output-bias gradient = column sums of d x_hat rides in the first BatchNorm-backward launch of the step instead of being a
// launch of its own at the head of the backward pass; 47 short workgroups beside 375 long ones)
// (two workgroups per CU, 120 VGPRs: limited to one per CU by 84 KB of unused LDS the step takes 10 us longer,
//  profiles/r03_ab_bn_bwd_one_per_cu_rejected.log -- the opposite of the forward kernel)
template <int R, int CQ>
__global__ __launch_bounds__(128 * CQ) void bn_act_bwd4_kernel(BnBwdGroup g, float p_drop, float slope, const uint64_t* rng,
                                                               ColsumGroup cs, int cs_begin, PfRanges pf, int pf_begin) {
    if ((int)blockIdx.x >= pf_begin) {
        prefetch_block(pf, (int)blockIdx.x - pf_begin, (int)gridDim.x - pf_begin);
        return;
    }
    if ((int)blockIdx.x >= cs_begin) {
        __shared__ float4 csh[32][17];
        float* o;
        colsum_block(cs, (int)blockIdx.x - cs_begin, csh, &o);
        return;
    }
    __shared__ float shraw[2 * CQ * 4 * CQ];
    __shared__ float sh2raw[2 * CQ * 8 * CQ];
    float (*sh)[4 * CQ] = reinterpret_cast<float (*)[4 * CQ]>(shraw);
    float (*sh2)[8 * CQ] = reinterpret_cast<float (*)[8 * CQ]>(sh2raw);
    __shared__ __attribute__((aligned(16))) unsigned short tl[CQ == 4 ? BN_CW * (128 * R + 2) : 8];
    int bb[JAMIE_MAX_GROUP];                    // (up front and by value: see the forward kernel)
#pragma unroll
    for (int i = 0; i < JAMIE_MAX_GROUP; ++i) bb[i] = g.p[i].blk_begin;
    const int cnt = g.count;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < JAMIE_MAX_GROUP; ++i) pi = (i < cnt && (int)blockIdx.x >= bb[i]) ? i : pi;
    const BnBwdDev P = g.p[pi];
    const int tid = threadIdx.x, cq = tid & (CQ - 1), rp = tid / CQ;
    const int col0 = bn_strip((int)blockIdx.x - P.blk_begin, P.N, 4 * CQ) * (4 * CQ), col = col0 + 4 * cq;
    const bool cok = col < P.N;
    const int B = P.B, N = P.N, nslab = P.nslab;
    float mean[4] = {0.f, 0.f, 0.f, 0.f}, invstd[4] = {0.f, 0.f, 0.f, 0.f}, ga[4] = {0.f, 0.f, 0.f, 0.f}, be[4] = {0.f, 0.f, 0.f, 0.f};
    if (cok) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { mean[e] = P.smean[col + e]; invstd[e] = P.sinvstd[col + e]; ga[e] = P.gamma[col + e]; be[e] = P.beta[col + e]; }
    }
    const bool drop = p_drop > 0.f;
    const float keep_scale = drop ? 1.f / (1.f - p_drop) : 1.f;
    const uint32_t thr = drop_threshold16(p_drop);
    // panel layout of the two fp32 inputs (bn_fwd_strip.h): this strip's rows are one contiguous block of each
    const bool pan = P.panel != 0;
    constexpr unsigned PW = JAMIE_PANEL, PB = 4u * JAMIE_PANEL;
    const unsigned row_bytes = pan ? PB : (unsigned)N * 4u, slab_bytes = (unsigned)(P.slab_stride * 4);
    const unsigned one_slab = pan ? (unsigned)((N + PW - 1) / PW) * (unsigned)B * PB : (unsigned)B * (unsigned)N * 4u;
    const __amdgpu_buffer_rsrc_t d_rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)P.da, 0, (int)((unsigned)(nslab - 1) * slab_bytes + one_slab), 0x00020000);
    const __amdgpu_buffer_rsrc_t h_rs = __builtin_amdgcn_make_buffer_rsrc((void*)P.h, 0, (int)one_slab, 0x00020000);
    const __amdgpu_buffer_rsrc_t m_rs = __builtin_amdgcn_make_buffer_rsrc((void*)P.mask, 0, P.mask ? B * N : 0, 0x00020000);
    const unsigned coff = pan ? ((unsigned)col / PW) * ((unsigned)B * PB) + ((unsigned)col % PW) * 4u : (unsigned)col * 4u;
    unsigned roff[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int row = rp + j * BN4_RP;
        roff[j] = (row < B && cok) ? (unsigned)row * row_bytes + coff : BN_OOB;
    }
    float4 dyv[R], xnv[R];
#pragma unroll
    for (int j = 0; j < R; ++j) xnv[j] = buf_f32x4<JAMIE_BN_LD_AUX>(h_rs, roff[j]);
#pragma unroll
    for (int j = 0; j < R; ++j) dyv[j] = buf_f32x4<JAMIE_BN_LD_AUX>(d_rs, roff[j]);
    unsigned mk[R];
    if (drop && P.mask) {
#pragma unroll
        for (int j = 0; j < R; ++j)       // (explicit masks: row-major [B, N] bytes)
            mk[j] = buf_u32(m_rs, roff[j] == BN_OOB ? BN_OOB : (unsigned)(rp + j * BN4_RP) * (unsigned)N + (unsigned)col);
    }
    unsigned keepbits = 0xFFFFFFFFu;              // Philox keep words under the loads in flight (see the forward kernel)
    if (drop) {
        keepbits = 0u;
#pragma unroll
        for (int j = 0; j < R; ++j) {
            if (P.mask) {
#pragma unroll
                for (int e = 0; e < 4; ++e) keepbits |= (((mk[j] >> (8 * e)) & 0xFFu) != 0 ? 1u : 0u) << (4 * j + e);
            } else if ((j & 1) == 0) {
                const Philox4 r = drop_rand4(rng, P.rng_stream, col, rp + j * BN4_RP);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    keepbits |= ((r.v[e] & 0xFFFFu) >= thr ? 1u : 0u) << (4 * j + e);
                    keepbits |= ((r.v[e] >> 16) >= thr ? 1u : 0u) << (4 * (j + 1) + e);
                }
            }
        }
    }
    for (int s = 1; s < nslab; ++s) {
        float4 a[R];
#pragma unroll
        for (int j = 0; j < R; ++j) a[j] = buf_f32x4<JAMIE_BN_LD_AUX>(d_rs, roff[j] == BN_OOB ? BN_OOB : roff[j] + (unsigned)s * slab_bytes);
#pragma unroll
        for (int j = 0; j < R; ++j) { dyv[j].x += a[j].x; dyv[j].y += a[j].y; dyv[j].z += a[j].z; dyv[j].w += a[j].w; }
    }
    float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int row = rp + j * BN4_RP;
        const bool ok = cok && row < B;
        float d[4] = {dyv[j].x, dyv[j].y, dyv[j].z, dyv[j].w}, x[4] = {xnv[j].x, xnv[j].y, xnv[j].z, xnv[j].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float xn = ok ? (x[e] - mean[e]) * invstd[e] : 0.f;
            float dd = d[e];
            if (drop) dd = ((keepbits >> (4 * j + e)) & 1u) ? dd * keep_scale : 0.f;
            const float y = xn * ga[e] + be[e];
            dd = ok ? (y > 0.f ? dd : dd * slope) : 0.f;
            x[e] = xn; d[e] = dd;
        }
        xnv[j] = make_float4(x[0], x[1], x[2], x[3]);
        dyv[j] = make_float4(d[0], d[1], d[2], d[3]);
        s1.x += d[0]; s1.y += d[1]; s1.z += d[2]; s1.w += d[3];
        s2.x += d[0] * x[0]; s2.y += d[1] * x[1]; s2.z += d[2] * x[2]; s2.w += d[3] * x[3];
    }
    col_reduce4x2<CQ>(s1, s2, sh2, tid);
    const float4 dbeta = s1, dgamma = s2;
    const float invB = 1.f / (float)B;
    const float k1[4] = {dbeta.x * invB, dbeta.y * invB, dbeta.z * invB, dbeta.w * invB};
    const float k2[4] = {dgamma.x * invB, dgamma.y * invB, dgamma.z * invB, dgamma.w * invB};
    float4 s3 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int row = rp + j * BN4_RP;
        float dh[4] = {0.f, 0.f, 0.f, 0.f};
        if (cok && row < B) {
            const float d[4] = {dyv[j].x, dyv[j].y, dyv[j].z, dyv[j].w}, x[4] = {xnv[j].x, xnv[j].y, xnv[j].z, xnv[j].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) dh[e] = ga[e] * invstd[e] * (d[e] - k1[e] - x[e] * k2[e]);
            if (!P.skip_f32) *reinterpret_cast<float4*>(P.da + (long long)row * N + col) = make_float4(dh[0], dh[1], dh[2], dh[3]);
            s3.x += dh[0]; s3.y += dh[1]; s3.z += dh[2]; s3.w += dh[3];
        }
        dyv[j] = make_float4(dh[0], dh[1], dh[2], dh[3]);
    }
    // the bias-gradient column sums come BEFORE the bf16 stores: col_reduce4's barriers are `__syncthreads()`, which drain
    // vmcnt(0) -- behind the stores they made every wave wait for its stores to be acknowledged
    const float4 dbias = col_reduce4<CQ>(s3, sh, tid);
    if (P.dh_bf || P.dhT_bf) strip_out_bf16x4<R, CQ>(dyv, P.dh_bf, P.dhT_bf, tl, B, N, col0, tid, cok);
    if (cok && rp == 0) {
        const float dg[4] = {dgamma.x, dgamma.y, dgamma.z, dgamma.w}, db[4] = {dbeta.x, dbeta.y, dbeta.z, dbeta.w};
        const float dl[4] = {dbias.x, dbias.y, dbias.z, dbias.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (P.accumulate) {
                P.dgamma[col + e] += dg[e];
                P.dbeta[col + e] += db[e];
                if (P.dbias) P.dbias[col + e] += dl[e];
            } else {
                P.dgamma[col + e] = dg[e];
                P.dbeta[col + e] = db[e];
                if (P.dbias) P.dbias[col + e] = dl[e];
            }
        }
    }
}

static int bn_act_fwd_impl(const jamie_bnact_fwd_problem* pr, int count, float p_drop, float momentum, float eps, float slope,
                           const uint64_t* rng, const void* const* pf, const long long* pf_bytes, int n_pf, void* stream);
static int pf_fill(const void* const* pf, const long long* pf_bytes, int n_pf, PfRanges& r) {
    memset(&r, 0, sizeof(r));
    JAMIE_ARG(n_pf >= 0 && n_pf <= BN_PF_MAX && (n_pf == 0 || (pf && pf_bytes)), "0 <= prefetch ranges <= 8");
    for (int i = 0; i < n_pf; ++i) {
        JAMIE_ARG(pf_bytes[i] >= 0 && (pf_bytes[i] == 0 || (pf[i] && (uintptr_t)pf[i] % 16 == 0)), "prefetch range: 16-byte aligned");
        if (pf_bytes[i] > 0) { r.p[r.n] = (const char*)pf[i]; r.bytes[r.n] = pf_bytes[i]; ++r.n; }
    }
    return 0;
}

extern "C" int jamie_bn_act_fwd(const jamie_bnact_fwd_problem* pr, int count, float p_drop, float momentum,
                                float eps, float slope, const uint64_t* rng, void* stream) {
    return bn_act_fwd_impl(pr, count, p_drop, momentum, eps, slope, rng, nullptr, nullptr, 0, stream);
}

extern "C" int jamie_bn_act_fwd_pf(const jamie_bnact_fwd_problem* pr, int count, float p_drop, float momentum, float eps,
                                   float slope, const uint64_t* rng, const void* const* prefetch, const long long* prefetch_bytes,
                                   int n_prefetch, void* stream) {
    return bn_act_fwd_impl(pr, count, p_drop, momentum, eps, slope, rng, prefetch, prefetch_bytes, n_prefetch, stream);
}

// Strip width of the float4 kernels: 16 columns on 512 threads (CQ = 4 column quads).  8 columns on 256 threads (CQ = 2) was
// built in round 3 to balance the chip -- at config 2 the 2d-wide layers have 375 strips for 256 CUs: half of the CUs carry two
// workgroups, the others one -- and measured: the step 595 -> 639 us (profiles/r03_ab_bn_strip_width_rejected.log): a
// wave-instruction then covers 32 rows x 32 bytes instead of 16 rows x 64, every 128-byte line is fetched by four workgroups
// instead of two, and the address path, not the balance, sets these kernels' time.  32 columns on 1024 threads (CQ = 8: whole
// lines per row, 188 workgroups): 592 -> 601 us.  (The transposed bf16 copies are laid
// out for 16-column strips and always take CQ = 4).
// Forward, round 3: the forward kernel runs ONE workgroup per CU (130 VGPRs), so 16-column strips take ceil(strips / 256) rounds of
// ~9 us; a 32-column strip costs 1.6 x as much but halves the count.  Where that saves rounds -- the 2d-wide layers of config 2:
// 375 strips = two rounds against 188 = one -- the 32-column kernel is taken: 20.9 -> 17.3 us per launch, while the d-wide layers
// (188 strips: one round either way) keep 16 columns (12.6 / 11.1 us against 16.1 / 13.3; profiles/r03_by_grid_bn_fwd_cq8.txt).
// Backward (two workgroups per CU): 16 columns always (32: +6 us per step).
// Panel layout, round 5, REJECTED variant (-DJAMIE_PANEL=8 -DJAMIE_BN_PANEL_CQ2=1): with panels of 8 columns a strip of 8 columns
// (CQ = 2, 256 threads) is one contiguous block per slab too, three such workgroups fit a CU (12 waves at 130 VGPRs), and config
// 2's 2d-wide layers are then 750 strips = at most 3 per CU = 24 columns on the busiest CU against 32 now.  Measured: the step
// 546.8 -> 574.3 us (profiles/r05_ab_panel8_cq2_rejected.log): the bf16 activations leave in 16-byte row segments and three
// workgroups' load bursts share one CU's memory pipe; the balance is not what binds these launches.
#ifndef JAMIE_BN_PANEL_CQ2
#define JAMIE_BN_PANEL_CQ2 0
#endif
static int bn_pick_cq(long long strips16, bool needs16, bool fwd, bool panel = false, long long cols = 0) {
#ifdef JAMIE_EXPERIMENTS      // (the experiments build's tests compare against 16-column strips: JAMIE_BN_CQ=4)
    const char* e = getenv(fwd ? "JAMIE_BN_CQ_FWD" : "JAMIE_BN_CQ_BWD");
    if (!e) e = getenv("JAMIE_BN_CQ");
    if (e) {
        const int v = atoi(e);
        return (!needs16 && (v == 2 || v == 8)) ? v : 4;
    }
#endif
    if (needs16) return 4;
    if (panel && JAMIE_BN_PANEL_CQ2 && JAMIE_PANEL == 8 && cols > 0) {
        const long long u8 = (cols + 7) / 8, u16 = (cols + 15) / 16;
        const long long per8 = (u8 + 255) / 256, per16 = (u16 + 255) / 256;       // strips on the busiest CU
        if (per8 <= 3 && per8 * 8 < per16 * 16) return 2;
    }
    if (!fwd) return 4;
    const long long r16 = (strips16 + 255) / 256, r32 = ((strips16 + 1) / 2 + 255) / 256;
    return (double)r32 * 1.6 < (double)r16 ? 8 : 4;
}

static int bn_act_fwd_impl(const jamie_bnact_fwd_problem* pr, int count, float p_drop, float momentum, float eps, float slope,
                           const uint64_t* rng, const void* const* pf, const long long* pf_bytes, int n_pf, void* stream) {
    PfRanges pfr;
    { const int rc = pf_fill(pf, pf_bytes, n_pf, pfr); if (rc) return rc; }
    JAMIE_ARG(pr && count >= 1 && count <= JAMIE_MAX_GROUP, "1 <= count <= JAMIE_MAX_GROUP");
    JAMIE_ARG(p_drop >= 0.f && p_drop < 1.f, "0 <= p < 1");
    BnFwdGroup g;
    memset(&g, 0, sizeof(g));
    g.count = count;
    int blocks = 0, maxB = 0;
    bool need_rng = false, wide = true, any_bf = false, any_panel = false;
    for (int i = 0; i < count; ++i) {
        const jamie_bnact_fwd_problem& s = pr[i];
        JAMIE_ARG(s.h && s.gamma && s.beta && s.running_mean && s.running_var && s.save_mean && s.save_invstd,
                  "null pointer");
        JAMIE_ARG(s.B >= 1 && s.N >= 1 && s.nslab >= 1, "B, N, nslab >= 1");
        JAMIE_ARG(s.nslab == 1 || s.slab_stride >= (long long)s.B * s.N, "slab_stride too small");
        JAMIE_ARG(((long long)(s.nslab - 1) * s.slab_stride + (long long)s.B * s.N) * 4 < 0xFFFFFFF0LL,
                  "activation slabs must stay below 4 GiB");
        BnFwdDev& d = g.p[i];
        d.h = s.h; d.gamma = s.gamma; d.beta = s.beta; d.rmean = s.running_mean; d.rvar = s.running_var;
        d.smean = s.save_mean; d.sinvstd = s.save_invstd; d.out = s.out; d.mask = s.mask;
        d.out_bf = (unsigned short*)s.out_bf16; d.outT_bf = (unsigned short*)s.outT_bf16;
        JAMIE_ARG((!s.out_bf16 && !s.outT_bf16) || (s.B <= 8 * BN4_RP && s.B % 8 == 0),
                  "fused bf16 outputs need B <= 1024 and B % 8 == 0");
        if (s.out_bf16 || s.outT_bf16) any_bf = true;
        JAMIE_ARG(s.out || s.out_bf16 || s.outT_bf16, "no output requested");
        d.slab_stride = s.slab_stride; d.nslab = s.nslab; d.B = s.B; d.N = s.N; d.rng_stream = s.rng_stream;
        d.panel = s.panel ? 1 : 0;
        if (s.panel) {
            any_panel = true;
            const long long one = (long long)((s.N + JAMIE_PANEL - 1) / JAMIE_PANEL) * JAMIE_PANEL * s.B;          // floats of one slab in panels
            JAMIE_ARG(s.nslab == 1 || s.slab_stride >= one, "panel layout: slab_stride >= ceil(N / JAMIE_PANEL) * JAMIE_PANEL * B");
            JAMIE_ARG(((long long)(s.nslab - 1) * s.slab_stride + one) * 4 < 0xFFFFFFF0LL, "activation slabs must stay below 4 GiB");
        }
        d.blk_begin = blocks;
        blocks += 8 * (((s.N + BN_CW - 1) / BN_CW + 7) / 8);
        if (s.B > maxB) maxB = s.B;
        if (!s.mask && p_drop > 0.f) need_rng = true;
        if (s.N % 4 || s.slab_stride % 4 || (uintptr_t)s.h % 16 || (uintptr_t)s.out % 16 || (uintptr_t)s.mask % 4 ||
            (uintptr_t)s.out_bf16 % 8)
            wide = false;
    }
    JAMIE_ARG(!need_rng || rng != nullptr, "rng state required when no explicit mask is given");
    hipStream_t st = (hipStream_t)stream;
    JAMIE_ARG(!any_bf || maxB <= BN_MAXR * BN_RP || wide, "fused bf16 outputs with 512 < B <= 1024 need the float4 path (N % 4 == 0, aligned)");
    JAMIE_ARG(!any_panel || (wide && maxB <= 8 * BN4_RP), "panel layout: float4 kernels only (N % 4 == 0, aligned, B <= 1024)");
    const int pfb = pfr.n > 0 ? bn_pf_blocks() : 0;        // (the float4 kernels carry the prefetch rider; the others ignore it)
    bool needs16 = false;
    for (int i = 0; i < count; ++i) needs16 = needs16 || pr[i].outT_bf16 != nullptr;
    long long cols = 0;
    for (int i = 0; i < count; ++i) cols += (pr[i].N + 7) / 8 * 8;
    int cq = (wide && maxB <= 8 * BN4_RP) ? bn_pick_cq(blocks, needs16, true, any_panel, cols) : 4;
    if (cq == 8 && maxB > BN4_MAXR * BN4_RP) cq = 4;
    if (cq == 2) {                 // 8-column strips: the workgroup ranges of the problems again
        blocks = 0;
        for (int i = 0; i < count; ++i) { g.p[i].blk_begin = blocks; blocks += 8 * (((pr[i].N + 7) / 8 + 7) / 8); }
    }
    if (cq == 8) {                 // 32-column strips on 1024 threads (B <= 512 only)
        blocks = 0;
        for (int i = 0; i < count; ++i) { g.p[i].blk_begin = blocks; blocks += 8 * (((pr[i].N + 31) / 32 + 7) / 8); }
    }
    if (wide && maxB <= BN4_MAXR * BN4_RP && cq == 8)
        hipLaunchKernelGGL((bn_act_fwd4_kernel<4, 8>), dim3(blocks + pfb), dim3(1024), 0, st, g, p_drop, momentum, eps, slope, rng, pfr, blocks);
    else if (wide && maxB <= BN4_MAXR * BN4_RP && cq == 2)
        hipLaunchKernelGGL((bn_act_fwd4_kernel<4, 2>), dim3(blocks + pfb), dim3(256), 0, st, g, p_drop, momentum, eps, slope, rng, pfr, blocks);
    else if (wide && maxB <= 8 * BN4_RP && cq == 2)
        hipLaunchKernelGGL((bn_act_fwd4_kernel<8, 2>), dim3(blocks + pfb), dim3(256), 0, st, g, p_drop, momentum, eps, slope, rng, pfr, blocks);
    else if (wide && maxB <= BN4_MAXR * BN4_RP)
        hipLaunchKernelGGL((bn_act_fwd4_kernel<4, 4>), dim3(blocks + pfb), dim3(512), 0, st, g, p_drop, momentum, eps, slope, rng, pfr, blocks);
    else if (wide && maxB <= 8 * BN4_RP)
        hipLaunchKernelGGL((bn_act_fwd4_kernel<8, 4>), dim3(blocks + pfb), dim3(512), 0, st, g, p_drop, momentum, eps, slope, rng, pfr, blocks);
    else if (maxB <= BN_MAXR * BN_RP)
        hipLaunchKernelGGL(bn_act_fwd_kernel<true>, dim3(blocks), dim3(256), 0, st, g, p_drop, momentum, eps, slope, rng);
    else
        hipLaunchKernelGGL(bn_act_fwd_kernel<false>, dim3(blocks), dim3(256), 0, st, g, p_drop, momentum, eps, slope, rng);
    return jamie_launch_status("jamie_bn_act_fwd");
}

static int bn_act_bwd_impl(const jamie_bnact_bwd_problem* pr, int count, float p_drop, float slope, const uint64_t* rng,
                           const jamie_colsum_problem* csp, int cs_count, void* stream, const void* const* pf = nullptr,
                           const long long* pf_bytes = nullptr, int n_pf = 0);

extern "C" int jamie_bn_act_bwd(const jamie_bnact_bwd_problem* pr, int count, float p_drop, float slope,
                                const uint64_t* rng, void* stream) {
    return bn_act_bwd_impl(pr, count, p_drop, slope, rng, nullptr, 0, stream);
}

extern "C" int jamie_bn_act_bwd_cs(const jamie_bnact_bwd_problem* pr, int count, float p_drop, float slope,
                                   const uint64_t* rng, const jamie_colsum_problem* colsums, int n_colsums, void* stream) {
    JAMIE_ARG(colsums && n_colsums >= 1 && n_colsums <= JAMIE_MAX_GROUP, "1 <= column-sum problems <= JAMIE_MAX_GROUP");
    return bn_act_bwd_impl(pr, count, p_drop, slope, rng, colsums, n_colsums, stream);
}

extern "C" int jamie_bn_act_bwd_pf(const jamie_bnact_bwd_problem* pr, int count, float p_drop, float slope, const uint64_t* rng,
                                   const jamie_colsum_problem* colsums, int n_colsums, const void* const* prefetch,
                                   const long long* prefetch_bytes, int n_prefetch, void* stream) {
    JAMIE_ARG(n_colsums >= 0 && n_colsums <= JAMIE_MAX_GROUP && (n_colsums == 0 || colsums), "0 <= column-sum problems <= JAMIE_MAX_GROUP");
    return bn_act_bwd_impl(pr, count, p_drop, slope, rng, colsums, n_colsums, stream, prefetch, prefetch_bytes, n_prefetch);
}

static int bn_act_bwd_impl(const jamie_bnact_bwd_problem* pr, int count, float p_drop, float slope, const uint64_t* rng,
                           const jamie_colsum_problem* csp, int cs_count, void* stream, const void* const* pf,
                           const long long* pf_bytes, int n_pf) {
    PfRanges pfr;
    { const int rc = pf_fill(pf, pf_bytes, n_pf, pfr); if (rc) return rc; }
    JAMIE_ARG(pr && count >= 1 && count <= JAMIE_MAX_GROUP, "1 <= count <= JAMIE_MAX_GROUP");
    JAMIE_ARG(p_drop >= 0.f && p_drop < 1.f, "0 <= p < 1");
    BnBwdGroup g;
    memset(&g, 0, sizeof(g));
    g.count = count;
    int blocks = 0, maxB = 0;
    bool need_rng = false, wide = true, any_bf = false, any_panel = false;
    for (int i = 0; i < count; ++i) {
        const jamie_bnact_bwd_problem& s = pr[i];
        JAMIE_ARG(s.da && s.h && s.gamma && s.beta && s.save_mean && s.save_invstd && s.dgamma && s.dbeta,
                  "null pointer");
        JAMIE_ARG(s.B >= 1 && s.N >= 1 && s.nslab >= 1, "B, N, nslab >= 1");
        JAMIE_ARG(s.nslab == 1 || s.slab_stride >= (long long)s.B * s.N, "slab_stride too small");
        JAMIE_ARG(((long long)(s.nslab - 1) * s.slab_stride + (long long)s.B * s.N) * 4 < 0xFFFFFFF0LL,
                  "activation slabs must stay below 4 GiB");
        BnBwdDev& d = g.p[i];
        d.da = s.da; d.h = s.h; d.gamma = s.gamma; d.beta = s.beta; d.smean = s.save_mean;
        d.sinvstd = s.save_invstd; d.dgamma = s.dgamma; d.dbeta = s.dbeta; d.dbias = s.dbias_lin; d.mask = s.mask;
        d.dh_bf = (unsigned short*)s.dh_bf16; d.dhT_bf = (unsigned short*)s.dhT_bf16; d.skip_f32 = s.skip_f32;
        JAMIE_ARG((!s.dh_bf16 && !s.dhT_bf16) || (s.B <= 8 * BN4_RP && s.B % 8 == 0),
                  "fused bf16 outputs need B <= 1024 and B % 8 == 0");
        if (s.dh_bf16 || s.dhT_bf16) any_bf = true;
        JAMIE_ARG(!s.skip_f32 || s.dh_bf16 || s.dhT_bf16, "skip_f32 without a bf16 output");
        d.slab_stride = s.slab_stride; d.nslab = s.nslab; d.B = s.B; d.N = s.N; d.rng_stream = s.rng_stream;
        d.accumulate = s.accumulate; d.blk_begin = blocks;
        d.panel = s.panel ? 1 : 0;
        if (s.panel) {
            any_panel = true;
            const long long one = (long long)((s.N + JAMIE_PANEL - 1) / JAMIE_PANEL) * JAMIE_PANEL * s.B;
            JAMIE_ARG(s.skip_f32, "panel layout: dh leaves as bf16 only (skip_f32)");
            JAMIE_ARG(s.nslab == 1 || s.slab_stride >= one, "panel layout: slab_stride >= ceil(N / JAMIE_PANEL) * JAMIE_PANEL * B");
            JAMIE_ARG(((long long)(s.nslab - 1) * s.slab_stride + one) * 4 < 0xFFFFFFF0LL, "activation slabs must stay below 4 GiB");
        }
        blocks += 8 * (((s.N + BN_CW - 1) / BN_CW + 7) / 8);
        if (s.B > maxB) maxB = s.B;
        if (!s.mask && p_drop > 0.f) need_rng = true;
        if (s.N % 4 || s.slab_stride % 4 || (uintptr_t)s.da % 16 || (uintptr_t)s.h % 16 || (uintptr_t)s.mask % 4 ||
            (uintptr_t)s.dh_bf16 % 8)
            wide = false;
    }
    JAMIE_ARG(!need_rng || rng != nullptr, "rng state required when no explicit mask is given");
    hipStream_t st = (hipStream_t)stream;
    JAMIE_ARG(!any_bf || maxB <= BN_MAXR * BN_RP || wide, "fused bf16 outputs with 512 < B <= 1024 need the float4 path (N % 4 == 0, aligned)");
    JAMIE_ARG(!any_panel || (wide && maxB <= 8 * BN4_RP), "panel layout: float4 kernels only (N % 4 == 0, aligned, B <= 1024)");
    ColsumGroup cs;
    memset(&cs, 0, sizeof(cs));
    int cs_blocks = 0;
    for (int i = 0; i < cs_count; ++i) {
        const jamie_colsum_problem& q = csp[i];
        JAMIE_ARG(q.X && q.out && q.M > 0 && q.N > 0 && q.ld >= q.N && q.nslab >= 1, "column-sum problem");
        ColsumDev& d = cs.p[i];
        d.X = q.X; d.out = q.out; d.slab_stride = q.slab_stride; d.M = q.M; d.N = q.N; d.ld = q.ld; d.nslab = q.nslab;
        d.accumulate = q.accumulate; d.blk_begin = cs_blocks;
        cs_blocks += (q.N + 63) / 64;
    }
    cs.count = cs_count;
    const bool wide4 = wide && maxB <= 8 * BN4_RP;
    const int extra = wide4 ? cs_blocks : 0;            // the float4 kernels take the column sums as extra workgroups
    const int pfb = pfr.n > 0 ? bn_pf_blocks() : 0;
    bool needs16 = false;
    for (int i = 0; i < count; ++i) needs16 = needs16 || pr[i].dhT_bf16 != nullptr;
    long long cols = 0;
    for (int i = 0; i < count; ++i) cols += (pr[i].N + 7) / 8 * 8;
    int cq = wide4 ? bn_pick_cq(blocks, needs16, false, any_panel, cols) : 4;
    if (cq == 8 && maxB > BN4_MAXR * BN4_RP) cq = 4;
    if (cq == 2) {
        blocks = 0;
        for (int i = 0; i < count; ++i) { g.p[i].blk_begin = blocks; blocks += 8 * (((pr[i].N + 7) / 8 + 7) / 8); }
    }
    if (cq == 8) {
        blocks = 0;
        for (int i = 0; i < count; ++i) { g.p[i].blk_begin = blocks; blocks += 8 * (((pr[i].N + 31) / 32 + 7) / 8); }
    }
    if (wide && maxB <= BN4_MAXR * BN4_RP && cq == 8)
        hipLaunchKernelGGL((bn_act_bwd4_kernel<4, 8>), dim3(blocks + extra + pfb), dim3(1024), 0, st, g, p_drop, slope, rng, cs, blocks,
                           pfr, blocks + extra);
    else if (wide && maxB <= BN4_MAXR * BN4_RP && cq == 2)
        hipLaunchKernelGGL((bn_act_bwd4_kernel<4, 2>), dim3(blocks + extra + pfb), dim3(256), 0, st, g, p_drop, slope, rng, cs, blocks,
                           pfr, blocks + extra);
    else if (wide4 && cq == 2)
        hipLaunchKernelGGL((bn_act_bwd4_kernel<8, 2>), dim3(blocks + extra + pfb), dim3(256), 0, st, g, p_drop, slope, rng, cs, blocks,
                           pfr, blocks + extra);
    else if (wide && maxB <= BN4_MAXR * BN4_RP)
        hipLaunchKernelGGL((bn_act_bwd4_kernel<4, 4>), dim3(blocks + extra + pfb), dim3(512), 0, st, g, p_drop, slope, rng, cs, blocks,
                           pfr, blocks + extra);
    else if (wide4)
        hipLaunchKernelGGL((bn_act_bwd4_kernel<8, 4>), dim3(blocks + extra + pfb), dim3(512), 0, st, g, p_drop, slope, rng, cs, blocks,
                           pfr, blocks + extra);
    else if (maxB <= BN_MAXR * BN_RP)
        hipLaunchKernelGGL(bn_act_bwd_kernel<true>, dim3(blocks), dim3(256), 0, st, g, p_drop, slope, rng);
    else
        hipLaunchKernelGGL(bn_act_bwd_kernel<false>, dim3(blocks), dim3(256), 0, st, g, p_drop, slope, rng);
    int rc = jamie_launch_status("jamie_bn_act_bwd");
    if (rc == 0 && cs_count > 0 && !wide4) rc = jamie_colsum_group(csp, cs_count, stream);     // (dword kernels: a launch of its own)
    return rc;
}

// Global-norm gradient clip + Adam on one flat fp32 buffer, gfx950.
//
// Replaces torch.nn.utils.clip_grad_norm_(params, 1) + torch.optim.Adam.step + zero_grad
// (reference jamie.py:481,739-741).  All parameters, gradients and both Adam moments live in four flat
// fp32 buffers (every tensor a 16-byte aligned view), so the update is two HBM-bound streaming kernels:
//   jamie_grad_sqnorm : per-block sum of g^2                      (reads 4n bytes)
//   jamie_clip_adam   : every block re-sums the <= 4096 partials, derives the clip coefficient and
//                       updates p, m, v                           (reads 16n, writes 12n bytes)
// 16 B per lane loads/stores, grid-stride over <= 2048 blocks (cdna_hip_programming.md Guideline 11).
#include "common.h"
#include "sampler.h"
#include "latent_final.h"
#include "colsum.h"
#include "cast_tile.h"
#include "range_norm.h"

__global__ __launch_bounds__(256) void grad_sqnorm_kernel(const float* __restrict__ g, long long n,
                                                          float* partials, uint64_t* state) {
    __shared__ float red[4];
    const long long n4 = n >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 v = g4[i];
        acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    if (blockIdx.x == 0) {
        const long long i = (n4 << 2) + threadIdx.x;
        if (i < n) acc += g[i] * g[i];
    }
    const float t = block_sum(acc, red);
    if (threadIdx.x == 0) {
        partials[blockIdx.x] = t;
        if (blockIdx.x == 0 && state) state[1] += 1;   // step counter (Adam's t, RNG step)
    }
}

// bf16 gradient (the reduced gradient of the data-parallel exchange, left in its bf16 message buffer): 8 values per lane
__device__ __forceinline__ float bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __uint_as_float(u & 0xFFFF0000u); }

__global__ __launch_bounds__(256) void grad_sqnorm_bf16_kernel(const unsigned short* __restrict__ g, long long n,
                                                               float* partials, uint64_t* state) {
    __shared__ float red[4];
    const long long n8 = n >> 3;
    const uint4* g8 = reinterpret_cast<const uint4*>(g);
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        const uint4 u = g8[i];
        const float a0 = bf16_lo(u.x), a1 = bf16_hi(u.x), a2 = bf16_lo(u.y), a3 = bf16_hi(u.y);
        const float a4 = bf16_lo(u.z), a5 = bf16_hi(u.z), a6 = bf16_lo(u.w), a7 = bf16_hi(u.w);
        acc += (a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3) + (a4 * a4 + a5 * a5 + a6 * a6 + a7 * a7);
    }
    if (blockIdx.x == 0) {
        const long long i = (n8 << 3) + threadIdx.x;
        if (i < n) { const float a = bf16_lo(g[i]); acc += a * a; }
    }
    const float t = block_sum(acc, red);
    if (threadIdx.x == 0) {
        partials[blockIdx.x] = t;
        if (blockIdx.x == 0 && state) state[1] += 1;
    }
}

// RIDE: EXTRA workgroups after the `n_stream` streaming ones do work of the NEXT step that would otherwise be launches of its
// own on the critical path, beside 256 streaming workgroups that run for ~190 us anyway:
//   * one workgroup (if smp.idx): the next batch's sampler (state[1] already holds the next step's number), and / or
//   * the next batch's row gather + bf16 cast (jamie_cast_transpose's 64 x 64 tiles; the batch buffers are free once the last dW
//     product of this step has run).  Sampler and gather never ride together here: the gather needs the sampler's output.
// Cache policy of the optimiser's streams (A/B switches: tools/ab.sh + JAMIE_LIB builds).  Round 3: the stores of the moments m, v
// (322 MB per step, next read a whole step later) go out NON-TEMPORALLY (ST_NT bit 0): through the caches they stayed behind as
// dirty lines whose write-back ran into the next step's first launches -- the forward GEMMs took 31.5 instead of 28.8 us each and
// the kernel itself 196 instead of 191 us: 638 -> 622 us per step on one box, interleaved (profiles/r03_ab_adam_cache_policy.log;
// the stand-alone microbenchmark, tools/bench_adam.py, had said "2 % slower" in round 1: what the stores cost shows in the
// kernels AFTER them).  The master weights' store too (ST_NT bit 1; in the first build hipcc had silently dropped the hint from
// the four scalar nt stores it merged): no difference (r03_ab_adam_p_nt.log), left at the default policy.  Loads (LD_NT bit 0:
// master weights, bit 1: the bf16 gradient): the GRADIENT non-temporal takes 9 us off the kernel and 12 off the step (178 -> 169,
// 610 -> 598 us: r03_ab_adam_ldnt.log; it is dead once read); the master weights non-temporal cost the kernel 9 us.  Bit 2: the fp32 gradient (fp32 compute mode, fp32 messages) non-temporal as well:
// config 2 in fp32 1506 -> 1495 us per step, config 5's dimensions 7287 -> 7243 (r03_ab_adam_g32_nt.log): LD_NT = 6.
// The bf16 weight copy non-temporal or not: no difference once the moments' stores are (W16_NT stays 1).  Starting the streams
// at the second layer so that the first layer's bf16 weights are written last (state[2]): +5 us, rejected
// (r03_ab_adam_rotate_rejected.log).
// (The A/B builds behind these measurements -- -DJAMIE_ADAM_{ST,LD,W16}_NT -- are gone; the adopted policies are written out below.)
template <int U, int T, bool RIDE>
__global__ __launch_bounds__(T) void clip_adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, long long n,
                                                        const float* partials, int n_partials,
                                                        const float* hyper, const uint64_t* state,
                                                        unsigned short* __restrict__ p_bf16,
                                                        const unsigned short* __restrict__ g_bf16, SampleArgs smp, CastGroup casts,
                                                        int n_stream) {
    if constexpr (RIDE) {
        if ((int)blockIdx.x >= n_stream) {
            if (smp.idx) {
                __shared__ SampleLds smp_lds;
                jamie_sample_block(smp_lds, smp.idx, smp.B, smp.N, smp.offset, smp.replace, state, smp.rng_stream, smp.step_add);
            } else {
                __shared__ float ctile[64][65];
                cast_tile_block(casts, (int)blockIdx.x - n_stream, ctile);
            }
            return;
        }
    }
    const long long n4 = n >> 2;
    float4* p4 = reinterpret_cast<float4*>(p);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
    const long long nwg = RIDE ? n_stream : gridDim.x;          // streaming workgroups
    // plain loads/stores: non-temporal variants measured 2 % slower here (tools/bench_adam.py: 4.83 vs 4.75 TB/s)
    // state[2] (optional, float4 units, < n / 4): the stream STARTS there and wraps around, so that the range in front of it -- the
    // first layer's parameters when the caller passes the offset of the second layer -- is updated LAST and its bf16 weight copy is
    // the freshest thing in the caches when the next step's first product reads it
    const long long rot4 = state[2] > 0 && state[2] < n4 ? state[2] : 0;
    float4 pp[U], mm[U], vv[U], gg[U];
    auto load_batch = [&](long long i0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            long long i = i0 + u * T;
            if (i < n4) {
                i += rot4;
                if (i >= n4) i -= n4;
                // the moments are read once per step: non-temporal loads (in the step -2..-4 us; the master weights and the
                // stores of all three measured no better streamed)
                pp[u] = p4[i];
                mm[u] = make_float4(__builtin_nontemporal_load(&m[4 * i]), __builtin_nontemporal_load(&m[4 * i + 1]),
                                    __builtin_nontemporal_load(&m[4 * i + 2]), __builtin_nontemporal_load(&m[4 * i + 3]));
                vv[u] = make_float4(__builtin_nontemporal_load(&v[4 * i]), __builtin_nontemporal_load(&v[4 * i + 1]),
                                    __builtin_nontemporal_load(&v[4 * i + 2]), __builtin_nontemporal_load(&v[4 * i + 3]));
                if (g_bf16) {      // reduced gradient read straight from the bf16 message buffer (no fp32 copy-back pass)
                    const unsigned long long q64 = __builtin_nontemporal_load(reinterpret_cast<const unsigned long long*>(g_bf16) + i);
                    const uint2 q = make_uint2((unsigned)q64, (unsigned)(q64 >> 32));
                    gg[u] = make_float4(bf16_lo(q.x), bf16_hi(q.x), bf16_lo(q.y), bf16_hi(q.y));
                } else {
                    // (the fp32 gradient: dead once read, like the bf16 one)
                    gg[u] = make_float4(__builtin_nontemporal_load(&g[4 * i]), __builtin_nontemporal_load(&g[4 * i + 1]),
                                        __builtin_nontemporal_load(&g[4 * i + 2]), __builtin_nontemporal_load(&g[4 * i + 3]));
                }
            }
        }
    };
    // the first batch of the streams is requested before the prologue below (norm, bias corrections): nothing in it depends on them
    long long i0 = (long long)blockIdx.x * (T * U) + threadIdx.x;
    if (i0 < n4) load_batch(i0);
    __shared__ float red[T / 64];
    // latency order: the first batch of partial sums, the hyper-parameters and the step counter are all requested up front; the
    // bias corrections (two double-precision pow: a few hundred instructions) are computed while the partial sums are in flight
    float s = 0.f;
    float t0[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t0[u] = (int)threadIdx.x + u * T < n_partials ? partials[threadIdx.x + u * T] : 0.f;
    const float lr = hyper[8], b1 = hyper[9], b2 = hyper[10], eps = hyper[11], max_norm = hyper[12];
    const float gscale = hyper[13];
    const double t = (double)state[1];
    const float bc1 = (float)(1.0 - pow((double)b1, t));
    const float bc2s = (float)sqrt(1.0 - pow((double)b2, t));
    const float step_size = lr / bc1;
#pragma unroll
    for (int u = 0; u < 8; ++u) s += t0[u];
    for (int i0 = threadIdx.x + 8 * T; i0 < n_partials; i0 += 8 * T) {      // eight loads in flight per round trip, added in index order
        float tt[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) tt[u] = i0 + u * T < n_partials ? partials[i0 + u * T] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) s += tt[u];
    }
    s = block_sum(s, red);
    const float total = sqrtf(s) * gscale;                 // norm of the (averaged) gradient
    const float coef = fminf(max_norm / (total + 1e-6f), 1.0f) * gscale;
    const float omb1 = 1.f - b1, omb2 = 1.f - b2;
    auto upd = [&](float& pp, float gg, float& mm, float& vv) {
        gg *= coef;
        mm = mm + (gg - mm) * omb1;                        // exp_avg.lerp_(grad, 1 - beta1)
        vv = vv * b2 + omb2 * gg * gg;                     // mul_(beta2).addcmul_(grad, grad, 1 - beta2)
        const float denom = sqrtf(vv) / bc2s + eps;
        pp -= step_size * (mm / denom);
    };
    for (; i0 < n4;) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            long long i = i0 + u * T;
            if (i >= n4) continue;
            i += rot4;
            if (i >= n4) i -= n4;
            upd(pp[u].x, gg[u].x, mm[u].x, vv[u].x);
            upd(pp[u].y, gg[u].y, mm[u].y, vv[u].y);
            upd(pp[u].z, gg[u].z, mm[u].z, vv[u].z);
            upd(pp[u].w, gg[u].w, mm[u].w, vv[u].w);
            // the moments stored non-temporally (whole 16-byte vectors), the master weights through the caches
            typedef float adam_f4 __attribute__((ext_vector_type(4)));
            p4[i] = pp[u];
            __builtin_nontemporal_store((adam_f4){mm[u].x, mm[u].y, mm[u].z, mm[u].w}, reinterpret_cast<adam_f4*>(m) + i);
            __builtin_nontemporal_store((adam_f4){vv[u].x, vv[u].y, vv[u].z, vv[u].w}, reinterpret_cast<adam_f4*>(v) + i);
            if (p_bf16) {   // bf16 copy of the updated master weights for the bf16-compute GEMMs (+2 B/parameter)
                const unsigned short b0 = __builtin_bit_cast(unsigned short, (__bf16)pp[u].x), b1 = __builtin_bit_cast(unsigned short, (__bf16)pp[u].y);
                const unsigned short b2 = __builtin_bit_cast(unsigned short, (__bf16)pp[u].z), b3 = __builtin_bit_cast(unsigned short, (__bf16)pp[u].w);
                // non-temporal: the copy is next read by the following step's GEMMs, ~1 GB of optimiser traffic later, and a
                // plain store of this eighth stream cost the kernel 10 us in the step (237 -> 227 us; the fp32 streams
                // measured no better with nt)
                const unsigned long long pk = (unsigned long long)((unsigned)b0 | ((unsigned)b1 << 16)) |
                                              ((unsigned long long)((unsigned)b2 | ((unsigned)b3 << 16)) << 32);
                __builtin_nontemporal_store(pk, reinterpret_cast<unsigned long long*>(p_bf16) + i);
            }
        }
        i0 += nwg * (T * U);
        if (i0 < n4) load_batch(i0);
    }
    if (blockIdx.x == 0) {
        const long long i = (n4 << 2) + threadIdx.x;
        if (i < n) {
            upd(p[i], g_bf16 ? bf16_lo(g_bf16[i]) : g[i], m[i], v[i]);
            if (p_bf16) p_bf16[i] = __builtin_bit_cast(unsigned short, (__bf16)p[i]);
        }
    }
}

// jamie_grad_sqnorm_ranges*: the range chunks and the finaliser of range_norm.h as a launch of their own (+ column-sum riders)
template <bool FIN>
__global__ __launch_bounds__(256) void grad_sqnorm_ranges_kernel(const float* __restrict__ g, unsigned short* __restrict__ g16,
                                                                 SqRanges r, float* partials, uint64_t* state, LatFinal fin,
                                                                 int n_range_blocks, ColsumGroup cs, int cs_begin) {
    if constexpr (FIN) {
        // further extra workgroups: column sums (the decoder's output-bias gradient = column sums of d x_hat), written into
        // the gradient (and its bf16 copy) together with their squares -- not in `r` either
        if ((int)blockIdx.x >= cs_begin) {
            __shared__ float4 csh[32][17];
            __shared__ float cred[4];
            float* o;
            const float t = colsum_block(cs, (int)blockIdx.x - cs_begin, csh, &o);
            if (o && g16) g16[o - g] = __builtin_bit_cast(unsigned short, (__bf16)t);
            const float q = block_sum(t * t, cred);
            if (threadIdx.x == 0) partials[blockIdx.x] = q;
            return;
        }
        // the extra workgroup: the deferred finalisation of the latent backward pass (losses, d sigma, head-bias gradients)
        // and the sum of squares of what it wrote -- those ranges are not in `r`
        __shared__ float fred[(256 / 64 + 1) * (SM_SLOTS + 2)];
        if ((int)blockIdx.x == n_range_blocks) {
            latent_m_finalise(fin, fred, &partials[n_range_blocks], g, g16);
            return;
        }
    }
    __shared__ float red[4];
    sqnorm_range_chunk(g, g16, r, (int)blockIdx.x, partials, state, red);
}

// Chunk length of a range list: JAMIE_SQ_CHUNK (4096 elements per workgroup) where that gives at most 128 chunks -- every
// BASELINE configuration in bf16 mode --, else the smallest multiple of it that does (fp32 mode at config 5's dimensions: the
// skinny head / latent matrices are ranges there, 1.5 M elements -> 12288 per workgroup)
static long long sq_chunk_len(const long long* lengths, int count) {
    for (long long c = JAMIE_SQ_CHUNK;; c += JAMIE_SQ_CHUNK) {
        long long nb = 0;
        for (int i = 0; i < count; ++i) nb += (lengths[i] + c - 1) / c;
        if (nb <= 128 || c >= (1LL << 30)) return c;
    }
}

static int fill_ranges(const long long* offsets, const long long* lengths, int count, bool g16, SqRanges* r, int* nb_out) {
    int nb = 0;
    JAMIE_ARG(count <= 128, "more than 128 ranges");
    const long long chunk = sq_chunk_len(lengths, count);
    for (int i = 0; i < count; ++i) {
        JAMIE_ARG(offsets[i] >= 0 && lengths[i] >= 0, "negative range");
        JAMIE_ARG(!g16 || offsets[i] % 4 == 0, "bf16 copies need range offsets that are multiples of 4");
        for (long long o = 0; o < lengths[i]; o += chunk) {
            JAMIE_ARG(nb < 128, "more than 128 chunks");
            r->off[nb] = offsets[i] + o;
            r->len[nb] = (int)(lengths[i] - o < chunk ? lengths[i] - o : chunk);
            ++nb;
        }
    }
    *nb_out = nb;
    return 0;
}

int jamie_range_ride_fill(const float* g, void* g16, const long long* offsets, const long long* lengths, int count, float* partials,
                          int n_partials, uint64_t* state, const jamie_latent_m* fin, RangeRide* rr, int* blocks) {
    JAMIE_ARG(g && offsets && lengths && partials && count >= 1 && rr && blocks, "null pointer / empty");
    memset(rr, 0, sizeof(*rr));
    int nb = 0;
    const int rc = fill_ranges(offsets, lengths, count, g16 != nullptr, &rr->r, &nb);
    if (rc) return rc;
    JAMIE_ARG(nb >= 1 && n_partials == nb + (fin ? 1 : 0), "n_partials must equal jamie_sqnorm_range_blocks() (+ 1 with a finaliser)");
    JAMIE_ARG(n_partials <= JAMIE_MAX_NORM_PARTIALS, "too many partial sums");
    rr->g = g; rr->g16 = (unsigned short*)g16; rr->partials = partials; rr->state = state; rr->n_range_blocks = nb;
    if (fin) {
        JAMIE_ARG(fin->defer_final && !fin->accumulate && fin->partials && fin->hyper && fin->losses && fin->dsigma,
                  "finaliser: a deferred, non-accumulating jamie_latent_m");
        JAMIE_ARG(fin->M * 2 * fin->L <= 256 * 64, "finaliser: too many head-bias columns");
        jamie_latent_m_fill_final(fin, &rr->fin);
        rr->has_fin = 1;
    }
    *blocks = nb + (fin ? 1 : 0);
    return 0;
}

static int sqnorm_ranges_impl(const float* g, void* g16, const long long* offsets, const long long* lengths, int count,
                              float* partials, int n_partials, uint64_t* state, void* stream, const jamie_latent_m* lat = nullptr,
                              const jamie_colsum_problem* csp = nullptr, int cs_count = 0) {
    JAMIE_ARG(g && offsets && lengths && partials && count >= 1, "null pointer / empty");
    SqRanges r;
    int nb = 0;
    const int frc = fill_ranges(offsets, lengths, count, g16 != nullptr, &r, &nb);
    if (frc) return frc;
    ColsumGroup cs;
    memset(&cs, 0, sizeof(cs));
    int cs_blocks = 0;
    if (lat && csp) {
        JAMIE_ARG(cs_count >= 1 && cs_count <= JAMIE_MAX_GROUP, "1 <= column-sum problems <= JAMIE_MAX_GROUP");
        cs.count = cs_count;
        for (int i = 0; i < cs_count; ++i) {
            const jamie_colsum_problem& q = csp[i];
            JAMIE_ARG(q.X && q.out && q.M > 0 && q.N > 0 && q.ld >= q.N && q.nslab >= 1 && !q.accumulate, "column-sum problem");
            ColsumDev& d = cs.p[i];
            d.X = q.X; d.out = q.out; d.slab_stride = q.slab_stride; d.M = q.M; d.N = q.N; d.ld = q.ld; d.nslab = q.nslab;
            d.accumulate = 0; d.blk_begin = cs_blocks;
            cs_blocks += (q.N + 63) / 64;
        }
    }
    JAMIE_ARG(nb >= 1 && n_partials == nb + (lat ? 1 : 0) + cs_blocks,
              "n_partials must equal jamie_sqnorm_range_blocks() (+ 1 with a finaliser, + ceil(N / 64) per column-sum problem)");
    JAMIE_ARG(n_partials <= JAMIE_MAX_NORM_PARTIALS, "too many partial sums");
    JAMIE_ARG(!g16 || ((uintptr_t)g16 % 8) == 0, "g_bf16 must be 8-byte aligned");
    LatFinal fin;
    memset(&fin, 0, sizeof(fin));
    if (lat) {
        JAMIE_ARG(lat->defer_final && lat->partials && lat->hyper && lat->losses && lat->dsigma && !lat->accumulate,
                  "finaliser: a deferred, non-accumulating jamie_latent_m");
        JAMIE_ARG(lat->M * 2 * lat->L <= 256 * 64, "finaliser: too many head-bias columns");
        jamie_latent_m_fill_final(lat, &fin);
        hipLaunchKernelGGL((grad_sqnorm_ranges_kernel<true>), dim3(nb + 1 + cs_blocks), dim3(256), 0, (hipStream_t)stream, g,
                           (unsigned short*)g16, r, partials, state, fin, nb, cs, nb + 1);
    } else {
        hipLaunchKernelGGL((grad_sqnorm_ranges_kernel<false>), dim3(nb), dim3(256), 0, (hipStream_t)stream, g, (unsigned short*)g16,
                           r, partials, state, fin, nb, cs, nb);
    }
    return jamie_launch_status("jamie_grad_sqnorm_ranges");
}

extern "C" int jamie_grad_sqnorm_ranges(const float* g, const long long* offsets, const long long* lengths, int count,
                                        float* partials, int n_partials, uint64_t* state, void* stream) {
    return sqnorm_ranges_impl(g, nullptr, offsets, lengths, count, partials, n_partials, state, stream);
}

extern "C" int jamie_grad_sqnorm_ranges_g16(const float* g, void* g_bf16, const long long* offsets, const long long* lengths,
                                            int count, float* partials, int n_partials, uint64_t* state, void* stream) {
    JAMIE_ARG(g_bf16 != nullptr, "null bf16 gradient buffer");
    return sqnorm_ranges_impl(g, g_bf16, offsets, lengths, count, partials, n_partials, state, stream);
}

extern "C" int jamie_grad_sqnorm_ranges_fin(const float* g, void* g_bf16, const long long* offsets, const long long* lengths,
                                            int count, float* partials, int n_partials, uint64_t* state,
                                            const jamie_latent_m* fin, const jamie_colsum_problem* colsums, int n_colsums,
                                            void* stream) {
    JAMIE_ARG(fin != nullptr, "null latent descriptor");
    return sqnorm_ranges_impl(g, g_bf16, offsets, lengths, count, partials, n_partials, state, stream, fin, colsums, n_colsums);
}

extern "C" int jamie_sqnorm_range_blocks(const long long* lengths, int count) {
    int nb = 0;
    const long long chunk = sq_chunk_len(lengths, count);
    for (int i = 0; i < count; ++i) nb += (int)((lengths[i] + chunk - 1) / chunk);
    return nb;
}

static int grid_for(long long n) {
    long long b = (n / 4 + 255) / 256;
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;
    return (int)b;
}

extern "C" int jamie_grad_sqnorm(const float* g, long long n, float* partials, int n_partials, uint64_t* state,
                                 void* stream) {
    JAMIE_ARG(g && partials && n > 0, "null pointer / empty");
    JAMIE_ARG(((uintptr_t)g % 16) == 0, "g must be 16-byte aligned");
    const int grid = grid_for(n);
    JAMIE_ARG(n_partials == grid, "n_partials must equal jamie_optim_blocks(n)");
    hipLaunchKernelGGL(grad_sqnorm_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, g, n, partials, state);
    return jamie_launch_status("jamie_grad_sqnorm");
}

// launch shape of clip + Adam (tools/sweep_adam.sh builds the alternatives): float4 per thread and array in flight, threads per
// workgroup, streaming workgroups
#ifndef JAMIE_ADAM_U
#define JAMIE_ADAM_U 1
#endif
#ifndef JAMIE_ADAM_T
#define JAMIE_ADAM_T 512
#endif
#ifndef JAMIE_ADAM_GRID
#define JAMIE_ADAM_GRID 256
#endif
static constexpr int ADAM_U = JAMIE_ADAM_U, ADAM_T = JAMIE_ADAM_T, ADAM_GRID = JAMIE_ADAM_GRID;

static int clip_adam_impl(float* p, const float* g, const void* g_bf16, float* m, float* v, long long n, const float* partials,
                          int n_partials, const float* hyper, const uint64_t* state, void* p_bf16, const jamie_sample_args* smp,
                          const jamie_cast_problem* casts, int n_casts, void* stream) {
    JAMIE_ARG(p && (g || g_bf16) && m && v && partials && hyper && state && n > 0, "null pointer / empty");
    JAMIE_ARG(((uintptr_t)p % 16) == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)m % 16) == 0 &&
                  ((uintptr_t)v % 16) == 0 && ((uintptr_t)g_bf16 % 8) == 0, "buffers must be 16-byte aligned (bf16 gradient: 8)");
    JAMIE_ARG(n_partials >= 1 && n_partials <= JAMIE_MAX_NORM_PARTIALS, "n_partials");
    JAMIE_ARG(p_bf16 == nullptr || ((uintptr_t)p_bf16 % 8) == 0, "p_bf16 must be 8-byte aligned");
    long long need = (n / 4 + ADAM_T - 1) / ADAM_T;
    const int grid = (int)(need < 1 ? 1 : (need > ADAM_GRID ? ADAM_GRID : need));
    // one workgroup per CU (fewer, longer streams keep more DRAM pages open: 2048 workgroups x 1 float4 4.8 TB/s), eight
    // waves each with one float4 per array in flight: 5.8 TB/s where four waves x two float4 reached 5.0-5.2 on the slower
    // boxes of the pool and 5.7 on the faster ones (tools/bench_adam.py; in the step 231 -> 209 us on a slow box)
    SampleArgs sa;
    CastGroup cg;
    memset(&sa, 0, sizeof(sa));
    memset(&cg, 0, sizeof(cg));
    const bool has_smp = smp && smp->idx, has_cast = casts && n_casts > 0;
    JAMIE_ARG(!(has_smp && has_cast), "the sampler and the gather cannot ride in the same launch (the gather reads the sampler's output)");
    if (has_smp) {
        JAMIE_ARG(smp->B > 0 && smp->N > 0 && (smp->replace || (smp->B <= smp->N && smp->B <= SMP_HASH / 2)),
                  "sampler: B <= N and B <= 2048 without replacement");
        JAMIE_ARG(smp->N + smp->offset <= 0x7fffffffLL, "sampler: indices must fit int32");
        sa.idx = smp->idx; sa.B = smp->B; sa.N = smp->N; sa.offset = smp->offset; sa.replace = smp->replace;
        sa.rng_stream = smp->rng_stream; sa.step_add = smp->step_add;
        hipLaunchKernelGGL((clip_adam_kernel<ADAM_U, ADAM_T, true>), dim3(grid + 1), dim3(ADAM_T), 0, (hipStream_t)stream, p, g, m, v, n,
                           partials, n_partials, hyper, state, (unsigned short*)p_bf16, (const unsigned short*)g_bf16, sa, cg, grid);
    } else if (has_cast) {
        int blocks = 0;
        const int rc = jamie_cast_fill_group(casts, n_casts, &cg, &blocks);
        if (rc) return rc;
        hipLaunchKernelGGL((clip_adam_kernel<ADAM_U, ADAM_T, true>), dim3(grid + blocks), dim3(ADAM_T), 0, (hipStream_t)stream, p, g, m, v, n,
                           partials, n_partials, hyper, state, (unsigned short*)p_bf16, (const unsigned short*)g_bf16, sa, cg, grid);
    } else {
        hipLaunchKernelGGL((clip_adam_kernel<ADAM_U, ADAM_T, false>), dim3(grid), dim3(ADAM_T), 0, (hipStream_t)stream, p, g, m, v, n,
                           partials, n_partials, hyper, state, (unsigned short*)p_bf16, (const unsigned short*)g_bf16, sa, cg, grid);
    }
    return jamie_launch_status("jamie_clip_adam");
}

extern "C" int jamie_clip_adam(float* p, const float* g, float* m, float* v, long long n, const float* partials,
                               int n_partials, const float* hyper, const uint64_t* state, void* p_bf16, void* stream) {
    JAMIE_ARG(g != nullptr, "null gradient");
    return clip_adam_impl(p, g, nullptr, m, v, n, partials, n_partials, hyper, state, p_bf16, nullptr, nullptr, 0, stream);
}

extern "C" int jamie_clip_adam_g16(float* p, const void* g_bf16, float* m, float* v, long long n, const float* partials,
                                   int n_partials, const float* hyper, const uint64_t* state, void* p_bf16, void* stream) {
    JAMIE_ARG(g_bf16 != nullptr, "null gradient");
    return clip_adam_impl(p, nullptr, g_bf16, m, v, n, partials, n_partials, hyper, state, p_bf16, nullptr, nullptr, 0, stream);
}

extern "C" int jamie_clip_adam_ride(float* p, const void* g, int g_is_bf16, float* m, float* v, long long n,
                                    const float* partials, int n_partials, const float* hyper, const uint64_t* state,
                                    void* p_bf16, const jamie_sample_args* sample, const jamie_cast_problem* casts, int n_casts,
                                    void* stream) {
    JAMIE_ARG(g != nullptr, "null gradient");
    return clip_adam_impl(p, g_is_bf16 ? nullptr : (const float*)g, g_is_bf16 ? g : nullptr, m, v, n, partials, n_partials, hyper,
                          state, p_bf16, sample, casts, n_casts, stream);
}

extern "C" int jamie_grad_sqnorm_bf16(const void* g_bf16, long long n, float* partials, int n_partials, uint64_t* state,
                                      void* stream) {
    JAMIE_ARG(g_bf16 && partials && n > 0, "null pointer / empty");
    JAMIE_ARG(((uintptr_t)g_bf16 % 16) == 0, "g must be 16-byte aligned");
    const int grid = grid_for(n);
    JAMIE_ARG(n_partials == grid, "n_partials must equal jamie_optim_blocks(n)");
    hipLaunchKernelGGL(grad_sqnorm_bf16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)g_bf16, n,
                       partials, state);
    return jamie_launch_status("jamie_grad_sqnorm_bf16");
}

extern "C" int jamie_optim_blocks(long long n) { return grid_for(n); }

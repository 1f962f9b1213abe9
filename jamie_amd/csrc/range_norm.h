// Sum of squares over a list of ranges of the gradient buffer (the parameters whose producers do not emit partial sums
// themselves: biases, BatchNorm affine parameters, sigma, the skinny head / latent matrices), one chunk of <= 4096 elements per
// workgroup, plus one optional extra workgroup for the deferred finalisation of the fused latent backward pass.  Shared by
// optim.hip (jamie_grad_sqnorm_ranges*: a launch of its own) and gemm_bf16.hip (jamie_gemm_bf16_ranges: the same work as EXTRA
// workgroups of the backward pass's last dW launch, after which every gradient exists).  Any workgroup size.
#pragma once
#include "common.h"
#include "latent_final.h"

#define JAMIE_SQ_CHUNK 4096
struct SqRanges { long long off[128]; int len[128]; };
struct RangeRide {
    const float* g; unsigned short* g16;        // flat gradient; optional bf16 copy of every range (same offsets)
    float* partials; uint64_t* state;           // partials[blk]; state[1] (the step counter) += 1 by block 0
    SqRanges r; int n_range_blocks;
    LatFinal fin; int has_fin;                  // block n_range_blocks: latent_m_finalise, its squares -> partials[n_range_blocks]
};

__device__ __forceinline__ void sqnorm_range_chunk(const float* __restrict__ g, unsigned short* __restrict__ g16, const SqRanges& r,
                                                   int blk, float* partials, uint64_t* state, float* red) {
    const int NT = blockDim.x;
    const float* p = g + r.off[blk];
    unsigned short* q = g16 ? g16 + r.off[blk] : nullptr;          // bf16 copy of the range (same offsets)
    const int n = r.len[blk], n4 = n >> 2;
    float acc = 0.f;
    auto bf = [](float x) { return __builtin_bit_cast(unsigned short, (__bf16)x); };
    if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
        for (int i = threadIdx.x; i < n4; i += NT) {
            const float4 v = reinterpret_cast<const float4*>(p)[i];
            acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
            if (q) *reinterpret_cast<uint2*>(q + 4 * i) = make_uint2((unsigned)bf(v.x) | ((unsigned)bf(v.y) << 16),
                                                                      (unsigned)bf(v.z) | ((unsigned)bf(v.w) << 16));
        }
        for (int i = (n4 << 2) + threadIdx.x; i < n; i += NT) { acc += p[i] * p[i]; if (q) q[i] = bf(p[i]); }
    } else {
        for (int i = threadIdx.x; i < n; i += NT) { acc += p[i] * p[i]; if (q) q[i] = bf(p[i]); }
    }
    const float t = block_sum(acc, red);
    if (threadIdx.x == 0) {
        partials[blk] = t;
        if (blk == 0 && state) state[1] += 1;
    }
}

// `red`: (blockDim.x / 64 + 1) * (SM_SLOTS + 2) floats of LDS
__device__ __forceinline__ void range_ride_block(const RangeRide& rr, int blk, float* red) {
    if (rr.has_fin && blk == rr.n_range_blocks) {
        latent_m_finalise(rr.fin, red, &rr.partials[rr.n_range_blocks], rr.g, rr.g16);
        return;
    }
    sqnorm_range_chunk(rr.g, rr.g16, rr.r, blk, rr.partials, rr.state, red);
}

// host (optim.hip): validates the arguments of jamie_grad_sqnorm_ranges_fin (without column sums) and fills `rr`;
// *blocks = workgroups the ride needs (n_range_blocks + 1 with a finaliser)
int jamie_range_ride_fill(const float* g, void* g16, const long long* offsets, const long long* lengths, int count, float* partials,
                          int n_partials, uint64_t* state, const jamie_latent_m* fin, RangeRide* rr, int* blocks);

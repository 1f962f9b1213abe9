// Shared device/host helpers for libjamie_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/jamie_hip.h"

#define JAMIE_WAVE 64
#define JAMIE_MAX_PARTIALS 4096
#define JAMIE_MAX_NORM_PARTIALS 32768     // partial sums of squares clip + Adam adds up (one per dW tile + the range chunks)

// ------------------------------------------------------------------------------------------------
// error reporting
// ------------------------------------------------------------------------------------------------
extern thread_local char g_jamie_err[512];

inline int jamie_fail(int code, const char* fmt, const char* a = "", long long b = 0, long long c = 0) {
    snprintf(g_jamie_err, sizeof(g_jamie_err), fmt, a, b, c);
    return code;
}

#define JAMIE_ARG(cond, msg)                                                                        \
    do {                                                                                            \
        if (!(cond)) return jamie_fail(-1, "%s: argument check failed: " msg " [%lld %lld]", __func__, \
                                       0, 0);                                                       \
    } while (0)

inline int jamie_launch_status(const char* who) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_jamie_err, sizeof(g_jamie_err), "%s: launch failed: %s", who, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Philox4x32-10 (counter-based; Salmon et al. 2011).  key = seed, counter = (index, stream, step).
// ------------------------------------------------------------------------------------------------
struct Philox4 {
    uint32_t v[4];
};

__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                 uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
        uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0;
        uint32_t n1 = lo1;
        uint32_t n2 = hi0 ^ c3 ^ k1;
        uint32_t n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += W0; k1 += W1;
    }
    Philox4 o;
    o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
    return o;
}

// rng[0] = seed, rng[1] = step.  `stream` separates layers / uses; `idx4` is the element index / 4.
__device__ __forceinline__ Philox4 jamie_rand4(const uint64_t* rng, uint32_t stream, uint64_t idx4) {
    uint64_t seed = rng[0], step = rng[1];
    return philox4x32_10((uint32_t)idx4, (uint32_t)(idx4 >> 32), stream, (uint32_t)step,
                         (uint32_t)seed, (uint32_t)(seed >> 32) ^ (uint32_t)(step >> 32));
}

// keep-mask for dropout probability p: keep iff u32 >= p * 2^32
__device__ __forceinline__ uint32_t jamie_drop_threshold(float p) {
    double t = (double)p * 4294967296.0;
    if (t <= 0.0) return 0u;
    if (t >= 4294967295.0) return 0xFFFFFFFFu;
    return (uint32_t)t;
}

__device__ __forceinline__ bool jamie_keep(const uint64_t* rng, uint32_t stream, uint64_t elem, uint32_t thr) {
    Philox4 r = jamie_rand4(rng, stream, elem >> 2);
    return r.v[elem & 3] >= thr;
}

// two N(0,1) from two u32 (Box-Muller)
__device__ __forceinline__ void jamie_box_muller(uint32_t a, uint32_t b, float& n0, float& n1) {
    float u1 = ((float)a + 1.0f) * 2.3283064365386963e-10f;  // (0,1]
    float u2 = (float)b * 2.3283064365386963e-10f;
    float r = sqrtf(-2.0f * __logf(u1));
    float s, c;
    __sincosf(6.283185307179586f * u2, &s, &c);
    n0 = r * c;
    n1 = r * s;
}

// ------------------------------------------------------------------------------------------------
// reductions (wave = 64)
// ------------------------------------------------------------------------------------------------
// wave-wide sum on the VALU's DPP lane permutations (no LDS traffic; `__shfl_xor` is a ds_bpermute per stage: with 16 waves
// reducing 13-18 values each, the shuffles alone took 7 us of a 20 us launch -- tools/stamp_latent.sh).  Quad butterfly
// (quad_perm), then row_half_mirror and row_mirror leave every lane of a 16-lane row with the row's sum; the four row sums
// are read with v_readlane and added on the scalar unit.  Fixed order: deterministic.  Result uniform across the wave.
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // lane ^ 1
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // lane ^ 2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
    const int x = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(x, 48));
    return (r0 + r1) + (r2 + r3);
}

// workgroup barrier that publishes LDS only: `__syncthreads()` also drains vmcnt(0), i.e. waits for global stores (and
// loads) that the barrier does not need
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
}

// (round 1's reductions -- ds_bpermute butterflies, fencing barriers -- cost 3.5 us in clip + Adam alone: profiles/r02_ab_dpp_reductions*.log)
__device__ __forceinline__ float wave_sum(float v) { return wave_sum_dpp(v); }
#define JAMIE_RED_BARRIER() lds_barrier()

// block-wide sum; `red` is >= (blockDim.x/64) floats of LDS; result valid in every thread.  Its barriers order LDS only.
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    JAMIE_RED_BARRIER();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    JAMIE_RED_BARRIER();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

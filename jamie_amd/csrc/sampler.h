// Device-side np.random.choice(N, B, replace) by ONE workgroup (any size that is a multiple of 64): shared by the
// stand-alone sampler launch (misc.hip) and by the optimiser launch, whose last workgroup draws the NEXT step's batch
// (optim.hip) -- a one-workgroup kernel of its own costs a 5 us launch on the step's critical path.
#pragma once
#include "common.h"

// ---- device-side np.random.choice(N, B, replace): one workgroup, deterministic given (seed, step) ----
// Without replacement, two exact methods (both symmetric under relabelling of values, so the result is a
// uniform B-subset in uniformly random order):
//   N <= 4096 : every candidate gets a random 32-bit key; a bitonic sort in LDS orders the candidates and
//               the first B are taken (ties broken by index).
//   N  > 4096 : rounds of "every unresolved slot draws; a value belongs to the claim with the lowest
//               (round, slot) priority, so claims of earlier rounds are never displaced; losers redraw".
//               B <= 2048 < N/2, so a draw succeeds with probability > 1/2 and 64 rounds leave a failure
//               probability below 2^-64 per slot.  Every wave leaves the loop after at most 64 rounds.
#define SMP_HASH 4096
struct SampleLds { long long key[SMP_HASH]; int owner[SMP_HASH]; int pending; };
// what to draw; `step_add`: added to the step counter rng[1] (a launch that runs BEFORE the norm kernel has advanced the counter
// draws the next step's batch with step_add = 1)
struct SampleArgs { int32_t* idx; int B; long long N; long long offset; int replace; int rng_stream; int step_add; };

// Without replacement and N > 4096 the slots are processed in chunks of 1024 (the stand-alone kernel's workgroup size), a
// thread taking the slots tid, tid + NT, ... of a chunk: the index stream does not depend on the workgroup size.
__device__ __forceinline__ void jamie_sample_block(SampleLds& L, int32_t* idx, int B, long long N, long long offset, int replace,
                                                   const uint64_t* rng_in, int rng_stream, int step_add = 0) {
    const uint64_t rng[2] = {rng_in[0], rng_in[1] + (uint64_t)step_add};
    long long (&key)[SMP_HASH] = L.key;
    int (&owner)[SMP_HASH] = L.owner;
    int& pending = L.pending;
    const int tid = threadIdx.x, NT = blockDim.x;
    if (replace) {
        for (int slot = tid; slot < B; slot += NT) {
            Philox4 r = jamie_rand4(rng, (uint32_t)rng_stream, (uint64_t)slot << 8);
            const uint64_t u = ((uint64_t)r.v[0] << 32) | r.v[1];
            idx[slot] = (int32_t)((long long)(u % (uint64_t)N) + offset);
        }
        return;
    }
    if (N <= SMP_HASH) {
        // sort-based exact subset: key = (random32 << 32) | candidate, padding = +inf
        for (int i = tid; i < SMP_HASH; i += NT) {
            if (i < N) {
                Philox4 r = jamie_rand4(rng, (uint32_t)rng_stream, (uint64_t)i << 8);
                key[i] = (long long)(((uint64_t)(r.v[0] >> 1) << 32) | (uint32_t)i);
            } else {
                key[i] = 0x7fffffffffffffffLL;
            }
        }
        __syncthreads();
        for (int k = 2; k <= SMP_HASH; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < SMP_HASH; i += NT) {
                    const int ixj = i ^ j;
                    if (ixj > i) {
                        const long long a = key[i], b = key[ixj];
                        const bool up = (i & k) == 0;
                        if ((a > b) == up) { key[i] = b; key[ixj] = a; }
                    }
                }
                __syncthreads();
            }
        }
        for (int slot = tid; slot < B; slot += NT) idx[slot] = (int32_t)((key[slot] & 0xffffffffLL) + offset);
        return;
    }
    for (int i = tid; i < SMP_HASH; i += NT) { key[i] = -1; owner[i] = 0x7fffffff; }
    __syncthreads();
    const int nq = 1024 / NT;                       // slots per thread and chunk (NT = 1024: one, as ever; NT >= 256)
    for (int base = 0; base < B; base += 1024) {   // chunks of 1024 slots share the table
        bool need[4];
        long long val[4] = {0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 4; ++q) need[q] = q < nq && base + tid + q * NT < B;
        for (unsigned round = 0; round < 64; ++round) {
            if (tid == 0) pending = 0;
            __syncthreads();
            int pos[4] = {-1, -1, -1, -1};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (!need[q]) continue;
                const int slot = base + tid + q * NT;
                const int prio = (int)((((unsigned)(base >> 10) * 64u + round) << 11) | (unsigned)slot);
                Philox4 r = jamie_rand4(rng, (uint32_t)rng_stream, ((uint64_t)slot << 8) | round);
                const uint64_t u = ((uint64_t)r.v[0] << 32) | r.v[1];
                val[q] = (long long)(u % (uint64_t)N);
                // open addressing keyed by value
                unsigned hsh = (unsigned)((uint64_t)val[q] * 0x9E3779B97F4A7C15ull >> 52) & (SMP_HASH - 1);
                for (int probe = 0; probe < SMP_HASH; ++probe) {
                    const long long prev = (long long)atomicCAS((unsigned long long*)&key[hsh],
                                                                (unsigned long long)-1LL, (unsigned long long)val[q]);
                    if (prev == -1 || prev == val[q]) { pos[q] = (int)hsh; break; }
                    hsh = (hsh + 1) & (SMP_HASH - 1);
                }
                if (pos[q] >= 0) atomicMin(&owner[pos[q]], prio);
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (!need[q]) continue;
                const int slot = base + tid + q * NT;
                const int prio = (int)((((unsigned)(base >> 10) * 64u + round) << 11) | (unsigned)slot);
                if (pos[q] >= 0 && owner[pos[q]] == prio) need[q] = false;
                else atomicAdd(&pending, 1);
            }
            __syncthreads();
            const int pend = pending;
            __syncthreads();
            if (pend == 0) break;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int slot = base + tid + q * NT;
            if (q < nq && slot < B) idx[slot] = (int32_t)(val[q] + offset);
        }
        __syncthreads();
    }
}

// Column sums of a [M, N] matrix (bias gradients of the Linear layers that no BatchNorm follows): the device function behind
// jamie_colsum_group (misc.hip) and behind the extra workgroups of the step's range-norm launch (optim.hip), where the
// decoder's output-bias gradient rides instead of costing a launch of its own at the head of the backward pass.
#pragma once
#include "common.h"

// ---- out[n] (+)= sum_m sum_slabs X[m,n]: 16 columns x 16 row phases per workgroup; up to 4 matrices per launch ----
struct ColsumDev { const float* X; float* out; long long slab_stride; int M, N, ld, nslab, accumulate, blk_begin; };
struct ColsumGroup { ColsumDev p[JAMIE_MAX_GROUP]; int count; };

// 64 columns per workgroup: 16 lanes x float4 across the columns (256 contiguous bytes per row), 16 row groups; a thread
// keeps 8 rows in flight.  (The first version read one dword per lane, 16 columns per workgroup: 11.5 us for the
// [512, 2000 + 1000] bias gradients of a step.)  Rows are added in a fixed order: deterministic.
// `blk`: block index within the group; returns (threads 0..63) the column sum this thread wrote, 0 elsewhere; `*oc_out` its column
// (any workgroup size >= 256: the first 256 threads do the work as 16 row groups x 16 column quads, so that the order of
//  the additions -- and with it every bit of the result -- does not depend on which launch the workgroup rides in)
__device__ __forceinline__ float colsum_block(const ColsumGroup& g, int blk, float4 (*sh)[17], float** out_ptr) {
    constexpr int NRG = 16;
    int pi = 0;
#pragma unroll
    for (int i = 1; i < JAMIE_MAX_GROUP; ++i)
        if (i < g.count && blk >= g.p[i].blk_begin) pi = i;
    const ColsumDev& P = g.p[pi];
    const int c4 = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int col = (blk - P.blk_begin) * 64 + c4 * 4;
    const bool vec = ((P.ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(P.X) & 15) == 0) && ((P.slab_stride & 3) == 0);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col < P.N && threadIdx.x < 256) {
        for (int s = 0; s < P.nslab; ++s) {
            const float* X = P.X + s * P.slab_stride + col;
            if (vec && col + 3 < P.N) {
                int m = rg;
                for (; m + 7 * NRG < P.M; m += 8 * NRG) {
                    float4 v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(X + (long long)(m + NRG * u) * P.ld);
#pragma unroll
                    for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
                }
                for (; m < P.M; m += NRG) {
                    const float4 v = *reinterpret_cast<const float4*>(X + (long long)m * P.ld);
                    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
                }
            } else {
                for (int m = rg; m < P.M; m += NRG) {
                    const float* r = X + (long long)m * P.ld;
                    acc.x += r[0];
                    if (col + 1 < P.N) acc.y += r[1];
                    if (col + 2 < P.N) acc.z += r[2];
                    if (col + 3 < P.N) acc.w += r[3];
                }
            }
        }
    }
    if (threadIdx.x < 256) sh[rg][c4] = acc;
    __syncthreads();
    *out_ptr = nullptr;
    if (threadIdx.x < 64) {
        const int c = threadIdx.x, oc = (blk - P.blk_begin) * 64 + c;
        if (oc < P.N) {
            float t = 0.f;
            for (int i = 0; i < NRG; ++i) t += reinterpret_cast<const float*>(&sh[i][c >> 2])[c & 3];
            t = P.accumulate ? P.out[oc] + t : t;
            P.out[oc] = t;
            *out_ptr = P.out + oc;
            return t;
        }
    }
    return 0.f;
}


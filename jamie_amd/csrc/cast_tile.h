// fp32 [R, C] (sum of slabs; optional row gather) -> fp32 / bf16 [R, C] and / or bf16 transposed [C, R] in 64 x 64 tiles: the device
// function behind jamie_cast_transpose (gemm_bf16.hip) and behind the extra workgroups of the optimiser launch, in which the NEXT
// step's batch gather rides (optim.hip).  Any workgroup size >= 256: the first 256 threads do the work.
#pragma once
#include "common.h"

#define CT_MAX 16
struct CastDev { const float* src; const unsigned short* src_bf; unsigned short* dst; unsigned short* dstT; const int32_t* rows; float* dst32; long long slab_stride; int R, C, ld, ldd, ldt, ld32, nslab, blk_begin, tiles_c; };
struct CastGroup { CastDev p[CT_MAX]; int count; };

// 64 x 64 tile per workgroup: float4 reads (a wave = 4 rows x 256 B), 8-byte bf16x4 row-major stores, the
// transposed copy through a padded LDS tile as 8-byte stores of 4 consecutive rows (16 lanes = 128 B).
__device__ __forceinline__ void cast_tile_block(const CastGroup& g, int blk, float (*tile)[65]) {
    int pi = 0;
    for (int i = 1; i < CT_MAX; ++i)
        if (i < g.count && blk >= g.p[i].blk_begin) pi = i;
    const CastDev& P = g.p[pi];
    const int b = blk - P.blk_begin;
    const bool act = threadIdx.x < 256;
    const int r0 = (b / P.tiles_c) * 64, c0 = (b % P.tiles_c) * 64;
    const int q = threadIdx.x & 15, rr0 = threadIdx.x >> 4;          // 16 column quads x 16 rows per pass
    const bool vec = P.src && (P.ld % 4 == 0) && (((uintptr_t)P.src & 15) == 0) && (P.slab_stride % 4 == 0);
    const bool vecd = P.dst && (P.ldd % 4 == 0) && (((uintptr_t)P.dst & 7) == 0);
    if (act) {
    // fp32 sources with 16-byte rows (the batch gather x = data[idx], slab sums): the loads of all four passes are issued
    // before the first store -- random rows of a matrix far larger than the caches are HBM misses, and a store between two
    // passes' loads (the pointers may alias) serialised four such round trips
    float4 pre[4];
    const bool prefetch = vec && !P.src_bf && P.nslab <= 4;
    if (prefetch) {
        long long gr[4];
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int r = r0 + rr0 + 16 * pass;
            gr[pass] = r < P.R ? (P.rows ? (long long)P.rows[r] : (long long)r) : 0;
        }
        float4 t[4][4];
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int r = r0 + rr0 + 16 * pass, c = c0 + 4 * q;
#pragma unroll
            for (int u = 0; u < 4; ++u)
                t[pass][u] = (r < P.R && c + 3 < P.C && u < P.nslab)
                                 ? *reinterpret_cast<const float4*>(P.src + u * P.slab_stride + gr[pass] * P.ld + c)
                                 : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            pre[pass] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < 4; ++u) { pre[pass].x += t[pass][u].x; pre[pass].y += t[pass][u].y; pre[pass].z += t[pass][u].z; pre[pass].w += t[pass][u].w; }
        }
    }
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int rr = rr0 + 16 * pass, r = r0 + rr, c = c0 + 4 * q;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (prefetch && r < P.R && c + 3 < P.C) {
            v = pre[pass];
        } else if (r < P.R && P.src_bf) {       // bf16 source (the copy the Adam kernel wrote): half the read traffic
            const unsigned short* sp = P.src_bf + (long long)r * P.ld + c;
            if ((P.ld % 4 == 0) && (((uintptr_t)P.src_bf & 7) == 0) && c + 3 < P.C) {
                const uint2 u = *reinterpret_cast<const uint2*>(sp);
                v.x = __uint_as_float(u.x << 16); v.y = __uint_as_float(u.x & 0xFFFF0000u);
                v.z = __uint_as_float(u.y << 16); v.w = __uint_as_float(u.y & 0xFFFF0000u);
            } else {
                if (c < P.C) v.x = __uint_as_float((unsigned)sp[0] << 16);
                if (c + 1 < P.C) v.y = __uint_as_float((unsigned)sp[1] << 16);
                if (c + 2 < P.C) v.z = __uint_as_float((unsigned)sp[2] << 16);
                if (c + 3 < P.C) v.w = __uint_as_float((unsigned)sp[3] << 16);
            }
        } else if (r < P.R) {
            const long long gr = P.rows ? (long long)P.rows[r] : (long long)r;      // row gather: x = data[idx] (jamie.py:583)
            for (int s = 0; s < P.nslab; ++s) {
                const float* sp = P.src + s * P.slab_stride + gr * P.ld + c;
                if (vec && c + 3 < P.C) {
                    const float4 u = *reinterpret_cast<const float4*>(sp);
                    v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
                } else {
                    if (c < P.C) v.x += sp[0];
                    if (c + 1 < P.C) v.y += sp[1];
                    if (c + 2 < P.C) v.z += sp[2];
                    if (c + 3 < P.C) v.w += sp[3];
                }
            }
        }
        tile[rr][4 * q] = v.x; tile[rr][4 * q + 1] = v.y; tile[rr][4 * q + 2] = v.z; tile[rr][4 * q + 3] = v.w;
        if (P.dst32 && r < P.R) {        // fp32 copy of the (gathered, slab-summed) rows
            float* dp = P.dst32 + (long long)r * P.ld32 + c;
            if ((P.ld32 % 4 == 0) && (((uintptr_t)P.dst32 & 15) == 0) && c + 3 < P.C) {
                *reinterpret_cast<float4*>(dp) = v;
            } else {
                if (c < P.C) dp[0] = v.x;
                if (c + 1 < P.C) dp[1] = v.y;
                if (c + 2 < P.C) dp[2] = v.z;
                if (c + 3 < P.C) dp[3] = v.w;
            }
        }
        if (P.dst && r < P.R) {
            const unsigned short b0 = __builtin_bit_cast(unsigned short, (__bf16)v.x), b1 = __builtin_bit_cast(unsigned short, (__bf16)v.y);
            const unsigned short b2 = __builtin_bit_cast(unsigned short, (__bf16)v.z), b3 = __builtin_bit_cast(unsigned short, (__bf16)v.w);
            unsigned short* dp = P.dst + (long long)r * P.ldd + c;
            if (vecd && c + 3 < P.C) {
                *reinterpret_cast<uint2*>(dp) = make_uint2((unsigned)b0 | ((unsigned)b1 << 16), (unsigned)b2 | ((unsigned)b3 << 16));
            } else {
                if (c < P.C) dp[0] = b0;
                if (c + 1 < P.C) dp[1] = b1;
                if (c + 2 < P.C) dp[2] = b2;
                if (c + 3 < P.C) dp[3] = b3;
            }
        }
    }
    }
    if (!P.dstT) return;
    __syncthreads();
    const bool vect = (P.ldt % 4 == 0) && (((uintptr_t)P.dstT & 7) == 0);
    if (act) {
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
        const int cc = rr0 + 16 * pass, c = c0 + cc, r = r0 + 4 * q;      // 4 consecutive rows of column c
        if (c >= P.C) continue;
        const unsigned short b0 = __builtin_bit_cast(unsigned short, (__bf16)tile[4 * q][cc]);
        const unsigned short b1 = __builtin_bit_cast(unsigned short, (__bf16)tile[4 * q + 1][cc]);
        const unsigned short b2 = __builtin_bit_cast(unsigned short, (__bf16)tile[4 * q + 2][cc]);
        const unsigned short b3 = __builtin_bit_cast(unsigned short, (__bf16)tile[4 * q + 3][cc]);
        unsigned short* dp = P.dstT + (long long)c * P.ldt + r;
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
}

// host: the device-side description of `count` cast problems and the number of 64 x 64 tiles (gemm_bf16.hip)
int jamie_cast_fill_group(const jamie_cast_problem* pr, int count, CastGroup* g, int* blocks);

// Shared helpers for the gfx950 kernels of libuavsal_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "uavsal_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define UAVSAL_NUM_XCD 8

static inline int uavsal_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

static inline bool uavsal_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// Blocks are dealt round-robin over the 8 XCDs (each with a private L2).  Map the
// hardware block id to a virtual id so that each XCD owns one contiguous range of
// virtual ids: neighbouring tiles (which share activation rows / halos / weights)
// then hit the same L2.  Bijective for any block count.  Speed only, never correctness.
__device__ __forceinline__ int xcd_virtual_block(int bid, int nblk) {
    const int q = nblk / UAVSAL_NUM_XCD, r = nblk % UAVSAL_NUM_XCD;
    const int xcd = bid % UAVSAL_NUM_XCD, slot = bid / UAVSAL_NUM_XCD;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

// Persistent-workgroup tile walk of the GEMMs.  Each XCD owns one contiguous range of output
// tiles; inside it the XCD's workgroups take tiles round-robin (tile, tile + stride, ...), so at
// any moment they sit on neighbouring tiles: the activation rows and weight rows in flight are
// a few M tiles' worth and stay in that XCD's 4 MB L2, instead of one M tile per workgroup.
struct uavsal_tile_walk { int tile, end, stride; };
#ifndef UAVSAL_TILE_WALK_STRIDED
#define UAVSAL_TILE_WALK_STRIDED 1
#endif
__device__ __forceinline__ uavsal_tile_walk xcd_tile_walk(int bid, int G, int nblk) {
    uavsal_tile_walk w;
#if UAVSAL_TILE_WALK_STRIDED
    const int q = G / UAVSAL_NUM_XCD, r = G % UAVSAL_NUM_XCD;
    const int xcd = bid % UAVSAL_NUM_XCD, slot = bid / UAVSAL_NUM_XCD;
    const int gx = q + (xcd < r ? 1 : 0);                         // workgroups on this XCD
    const int vb0 = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const int t0 = (int)(((long long)vb0 * nblk) / G);
    w.end = (int)(((long long)(vb0 + gx) * nblk) / G);
    w.tile = t0 + slot;
    w.stride = gx;
#else
    const int vb = xcd_virtual_block(bid, G);
    w.tile = (int)(((long long)vb * nblk) / G);
    w.end = (int)(((long long)(vb + 1) * nblk) / G);
    w.stride = 1;
#endif
    return w;
}

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == UAVSAL_ACT_RELU6) return fminf(fmaxf(v, 0.f), 6.f);
    if (act == UAVSAL_ACT_SIGMOID) return 1.f / (1.f + expf(-v));
    return v;
}

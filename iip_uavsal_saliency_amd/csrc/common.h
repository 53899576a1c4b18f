// Shared helpers for the gfx950 kernels of libuavsal_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
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

// Split shadow of four fp32 values (uavsal_hip.h, uavsal_conv_desc.a_hi): hi = fp16_rtz(16 x), lo = fp16_rtz(16 x - hi),
// packed as 4 halves each.  Round-toward-zero saturates at the largest finite half, so nothing becomes inf; the
// residual 16 x - hi is exact in fp32 whatever the rounding of hi.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void uavsal_split4_f16(f32x4 x, u32x2& hi, u32x2& lo) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef __fp16 h2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        f2 v = q ? (f2){x.z, x.w} : (f2){x.x, x.y};
        v = v * 16.0f;
        const h2 h = __builtin_amdgcn_cvt_pkrtz(v.x, v.y);
        const f2 r = v - (f2){(float)h.x, (float)h.y};
        const h2 l = __builtin_amdgcn_cvt_pkrtz(r.x, r.y);
        hi[q] = __builtin_bit_cast(unsigned, h);
        lo[q] = __builtin_bit_cast(unsigned, l);
    }
}

// store the split shadow of channels c..c+3 (c % 4 == 0) of one pixel: `row` points at the pixel's shadow row,
// layout [group of 32 channels][hi 32 halves | lo 32 halves] (uavsal_hip.h)
__device__ __forceinline__ void uavsal_store_split4(_Float16* row, int c, f32x4 v) {
    u32x2 sh, sl;
    uavsal_split4_f16(v, sh, sl);
    _Float16* p = row + (c >> 5) * 64 + (c & 31);
    *reinterpret_cast<u32x2*>(p) = sh;
    *reinterpret_cast<u32x2*>(p + 32) = sl;
}

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == UAVSAL_ACT_RELU6) return fminf(fmaxf(v, 0.f), 6.f);
    if (act == UAVSAL_ACT_SIGMOID) return 1.f / (1.f + expf(-v));
    return v;
}

// Dynamic LDS above the 64 KB default is an opt-in per kernel AND per device (one process may drive several GPUs: model(x.to("cuda:1"))).
// Placed in front of every launch of such a kernel (all of them in functions that return a status): after the first time on
// a device it is one hipGetDevice and a bit test.  The per-site mask is atomic (ctypes releases the GIL: two host threads may
// launch the same kernel), a device's bit is set only once the attribute call SUCCEEDED, and a failure is returned to the
// caller as the HIP error instead of surfacing later as a launch error.  Devices >= 64 are never memoised.
#define UAVSAL_LDS_OPTIN(kernel, bytes)                                                                              \
    do {                                                                                                             \
        static std::atomic<unsigned long long> uavsal_optin_done_{0ull};                                             \
        int uavsal_optin_dev_ = 0;                                                                                   \
        hipError_t uavsal_optin_e_ = hipGetDevice(&uavsal_optin_dev_);                                               \
        if (uavsal_optin_e_ != hipSuccess) return (int)uavsal_optin_e_;                                              \
        const bool uavsal_optin_memo_ = uavsal_optin_dev_ >= 0 && uavsal_optin_dev_ < 64;                            \
        if (!uavsal_optin_memo_ || !((uavsal_optin_done_.load(std::memory_order_acquire) >> uavsal_optin_dev_) & 1ull)) { \
            uavsal_optin_e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); \
            if (uavsal_optin_e_ != hipSuccess) return (int)uavsal_optin_e_;                                          \
            if (uavsal_optin_memo_) uavsal_optin_done_.fetch_or(1ull << uavsal_optin_dev_, std::memory_order_release); \
        }                                                                                                            \
    } while (0)

// A launch-time device query (resident workgroups of a kernel instance, CU count) memoised PER DEVICE: a process that
// drives several GPUs must not size the grids / stream-K plans of device 1 from device 0's answer.  `expr` is an int > 0.
template <typename F>
static inline int uavsal_per_device_memo(std::atomic<int>* slots, F&& fn) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return fn();
    int v = slots[dev].load(std::memory_order_acquire);
    if (v <= 0) {
        v = fn();
        slots[dev].store(v, std::memory_order_release);
    }
    return v;
}
#define UAVSAL_PER_DEVICE(expr) ([&] { static std::atomic<int> uavsal_slots_[64]; return uavsal_per_device_memo(uavsal_slots_, [&] { return (int)(expr); }); }())

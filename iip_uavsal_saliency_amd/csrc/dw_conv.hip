// Depthwise 3x3 convolution + folded BatchNorm + ReLU6 on NHWC fp32 activations.
// Replaces the groups=hidden_dim BasicConv2d inside dwBlock (reference model.py:92) and
// torchvision InvertedResidual's depthwise ConvBNReLU.
//
// HBM-bound (0.46 GMAC but 420.8 MiB per 360x640 frame), so the design
// is about bytes:
//   * NHWC with 4 channels (16 B) per lane: a wave reads 64 consecutive float4 = 1 KiB
//     of one pixel's channels per load instruction -- fully coalesced;
//   * each thread produces a TY x TX patch of output pixels for its 4 channels and walks
//     the input rows it needs once, keeping one row of (TX-1)*S+3 float4 in registers:
//     24 loads for 8 outputs at stride 1 instead of 72 -> L1/TA request rate stays well
//     below its limit while HBM sees each byte once;
//   * the 9 per-channel taps and the folded BN scale/bias live in registers;
//   * blockIdx is remapped so each XCD works on a contiguous slab of (image, row-band)
//     work: the halo rows shared by neighbouring patches are served by the same L2.
#include "common.h"
#include <stdlib.h>

namespace {

struct DwK {
    const float* in;
    const float* w9c;
    const float* scale;
    const float* bias;
    float* out;
    _Float16* out_split;                     // split shadow written INSTEAD of `out` (or null)
    int ldos;
    int ldi, ldo, H, W, Ho, Wo, C4, dil, act;
    int tiles_x, tiles_y, n_img;
    long long total;   // work items = n_img * tiles_y * tiles_x * C4
    int nblk;
    int dgc, dils[4];  // dgc > 0: channel c has dilation dils[c / dgc] (several dilated branches of one map in one launch)
    int rc_first[5];   // dw3x3_rowclass_kernel: first virtual block of every dilation group (and the total)
    int rc_q[4];       //   and the channel quads per slab of the group (16 / 32 / 64: the fewer rows a class has, the wider the slab)
};

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

template <int S, int TY, int TX>
__global__ __launch_bounds__(256) void dw3x3_kernel(const DwK p) {
    const int vb = xcd_virtual_block(blockIdx.x, p.nblk);
    const long long item = (long long)vb * 256 + threadIdx.x;
    if (item >= p.total) return;
    const int c4 = (int)(item % p.C4);
    long long t = item / p.C4;
    const int tx = (int)(t % p.tiles_x); t /= p.tiles_x;
    const int ty = (int)(t % p.tiles_y);
    const int n = (int)(t / p.tiles_y);
    const int c = c4 * 4;

    f32x4 wt[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wt[k] = ld4(p.w9c + (size_t)k * (p.C4 * 4) + c);
    const f32x4 sc = ld4(p.scale + c), bi = ld4(p.bias + c);

    constexpr int IW = (TX - 1) * S + 3;   // input columns per row of the patch
    constexpr int IH = (TY - 1) * S + 3;
    const int oy0 = ty * TY, ox0 = tx * TX;
    const int iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;   // dilation 1 path
    const float* inb = p.in + (size_t)n * p.H * p.W * p.ldi + c;

    f32x4 acc[TY][TX];
#pragma unroll
    for (int a = 0; a < TY; ++a)
#pragma unroll
        for (int b = 0; b < TX; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int r = 0; r < IH; ++r) {
        const int iy = iy0 + r;
        const bool rok = iy >= 0 && iy < p.H;
        f32x4 row[IW];
#pragma unroll
        for (int q = 0; q < IW; ++q) {
            const int ix = ix0 + q;
            const bool ok = rok && ix >= 0 && ix < p.W;
            row[q] = ok ? ld4(inb + ((size_t)iy * p.W + ix) * p.ldi) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int a = 0; a < TY; ++a) {
            const int ky = r - a * S;          // which kernel row this input row is for output row a
            if (ky < 0 || ky > 2) continue;
#pragma unroll
            for (int b = 0; b < TX; ++b)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) acc[a][b] += row[b * S + kx] * wt[ky * 3 + kx];
        }
    }

#pragma unroll
    for (int a = 0; a < TY; ++a) {
        const int oy = oy0 + a;
        if (oy >= p.Ho) continue;
#pragma unroll
        for (int b = 0; b < TX; ++b) {
            const int ox = ox0 + b;
            if (ox >= p.Wo) continue;
            f32x4 v = acc[a][b] * sc + bi;
            if (p.act == UAVSAL_ACT_RELU6) {
                v.x = fminf(fmaxf(v.x, 0.f), 6.f); v.y = fminf(fmaxf(v.y, 0.f), 6.f);
                v.z = fminf(fmaxf(v.z, 0.f), 6.f); v.w = fminf(fmaxf(v.w, 0.f), 6.f);
            }
            const size_t opix = ((size_t)n * p.Ho + oy) * p.Wo + ox;
            if (p.out_split) {   // the projection GEMM that consumes this tensor stages it pre-split (LDS-DMA)
                uavsal_store_split4(p.out_split + opix * p.ldos, c, v);
            } else {
                *reinterpret_cast<f32x4*>(p.out + opix * p.ldo + c) = v;
            }
        }
    }
}

// generic dilation (stride 1): one output pixel per thread, 9 bounds-checked taps.  Used by
// the three ASPP branches (dilation 6/12/18 on the 1/32-scale map, model.py:125-127), where
// most taps fall into the zero padding.
__global__ __launch_bounds__(256) void dw3x3_dilated_kernel(const DwK p) {
    const long long item = (long long)blockIdx.x * 256 + threadIdx.x;
    if (item >= p.total) return;
    const int c4 = (int)(item % p.C4);
    long long t = item / p.C4;
    const int ox = (int)(t % p.Wo); t /= p.Wo;
    const int oy = (int)(t % p.Ho);
    const int n = (int)(t / p.Ho);
    const int c = c4 * 4;
    const float* inb = p.in + (size_t)n * p.H * p.W * p.ldi + c;
    const int dil = p.dgc ? p.dils[c / p.dgc] : p.dil;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy + (ky - 1) * dil;
        if (iy < 0 || iy >= p.H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = ox + (kx - 1) * dil;
            if (ix < 0 || ix >= p.W) continue;
            acc += ld4(inb + ((size_t)iy * p.W + ix) * p.ldi) * ld4(p.w9c + (size_t)(ky * 3 + kx) * (p.C4 * 4) + c);
        }
    }
    f32x4 v = acc * ld4(p.scale + c) + ld4(p.bias + c);
    if (p.act == UAVSAL_ACT_RELU6) {
        v.x = fminf(fmaxf(v.x, 0.f), 6.f); v.y = fminf(fmaxf(v.y, 0.f), 6.f);
        v.z = fminf(fmaxf(v.z, 0.f), 6.f); v.w = fminf(fmaxf(v.w, 0.f), 6.f);
    }
    *reinterpret_cast<f32x4*>(p.out + (((size_t)n * p.Ho + oy) * p.Wo + ox) * p.ldo + c) = v;
}

// Small maps (the 1/32- and 1/16-scale levels: 12x20, 23x40 at 360x640): one workgroup stages the WHOLE
// map of one image for a slab of CB channels in LDS (<= 64 KB) with coalesced 16-byte loads, then every
// output reads its nine taps from LDS -- each input byte is fetched from L2/HBM exactly once whatever the
// dilation (the per-tap global loads of the generic dilated kernel made d=6 -- all nine taps inside the map
// -- twice as slow as d=18).  Stride 1, any dilation.  Thread i handles float4 item i of the slab
// (pixel = i / (CB/4), channel quad = i % (CB/4)); 256 % (CB/4) == 0, so a thread keeps one channel quad
// and its taps / BN constants stay in registers.
// Round 5: (a) the map is staged by LDS-DMA (global_load_lds_dwordx4: a wave request = 1 KB, lane-linear in LDS, which IS the
// [pixel][channel quad] order of `sm`; no VGPR round trip, every request of the workgroup in flight at once; lanes past the
// slab / the map fetch a zero page); (b) blocks are renumbered per XCD (xcd_virtual_block): a 16-channel slab reads and
// writes 64-byte half lines, and the slab holding the other half used to run on ANOTHER XCD (blocks are dealt round-robin),
// i.e. behind another L2 -- every line was fetched from and written to the memory side twice.  720x1280, 64 frames, 23x40 x
// 5760 channels (aspp.dw of BASELINE configs[4]): 0.37 of 8 TB/s before (profiles/r4_bench_720p_c4_t16.json).
__device__ __attribute__((aligned(16))) float g_dw_zero[4];
#ifdef UAVSAL_DW_STAMPS       // diagnostic build only: s_memrealtime (100 MHz) at the phase boundaries of the first 16384 workgroups
__device__ unsigned long long g_dw_stamps[4 * 16384];
#define DW_STAMP(k) { if (threadIdx.x == 0 && blockIdx.x < 16384) g_dw_stamps[blockIdx.x * 4 + (k)] = __builtin_amdgcn_s_memrealtime(); }
#else
#define DW_STAMP(k)
#endif

template <int CB, int NT = 256>
__global__ __launch_bounds__(NT) void dw3x3_map_lds_kernel(const DwK p) {
    extern __shared__ __attribute__((aligned(16))) float smap[];
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    constexpr int Q = CB / 4;
    const int slabs = (p.C4 * 4 + CB - 1) / CB;
    const int vb = xcd_virtual_block(blockIdx.x, gridDim.x);
    const int n = vb / slabs, c0 = (vb - n * slabs) * CB;
    static_assert(NT % Q == 0, "a thread keeps one channel quad");
    const int q = threadIdx.x % Q;
    const int c = c0 + q * 4;
    const bool cok = c < p.C4 * 4;                 // last slab may be narrower
    const int HW = p.H * p.W;
    const float* inb = p.in + (size_t)n * HW * p.ldi + c;
    f32x4* sm = reinterpret_cast<f32x4*>(smap);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    DW_STAMP(0)
    for (int i0 = 0; i0 < HW * Q; i0 += NT) {      // (the LDS image is padded to whole NT-item rounds: launch_map_lds)
        const int i = i0 + threadIdx.x;
        const int pix = i / Q;
        const float* src = (cok && pix < HW) ? inb + (size_t)pix * p.ldi : g_dw_zero;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smap + (size_t)(i0 + wave * 64) * 4), 16, 0, 0);
    }
    f32x4 wt[9];
    const int cc = cok ? c : 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) wt[k] = ld4(p.w9c + (size_t)k * (p.C4 * 4) + cc);
    const f32x4 sc = ld4(p.scale + cc), bi = ld4(p.bias + cc);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DW_STAMP(1)
    __syncthreads();
    DW_STAMP(2)
    if (!cok) return;
    const int dil = p.dgc ? p.dils[c0 / p.dgc] : p.dil;          // (a slab never straddles two groups: dgc % CB == 0)
    float* outb = p.out + (size_t)n * HW * p.ldo + c;
    for (int i = threadIdx.x; i < HW * Q; i += NT) {
        const int pix = i / Q;
        const int oy = pix / p.W, ox = pix - oy * p.W;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy + (ky - 1) * dil;
            if (iy < 0 || iy >= p.H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox + (kx - 1) * dil;
                if (ix < 0 || ix >= p.W) continue;
                acc += sm[(iy * p.W + ix) * Q + q] * wt[ky * 3 + kx];
            }
        }
        f32x4 v = acc * sc + bi;
        if (p.act == UAVSAL_ACT_RELU6) {
            v.x = fminf(fmaxf(v.x, 0.f), 6.f); v.y = fminf(fmaxf(v.y, 0.f), 6.f);
            v.z = fminf(fmaxf(v.z, 0.f), 6.f); v.w = fminf(fmaxf(v.w, 0.f), 6.f);
        }
        *reinterpret_cast<f32x4*>(outb + (size_t)pix * p.ldo) = v;
    }
#ifdef UAVSAL_DW_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DW_STAMP(3)
#endif
}

#ifdef UAVSAL_DW_STAMPS
extern "C" int uavsal_dw_stamps(unsigned long long* dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_dw_stamps), sizeof(unsigned long long) * 4 * 16384);
}
#endif

// Dilated maps, by ROW CLASS (round 5).  Output row oy of a dilation-d conv reads the input rows oy - d, oy, oy + d only: rows of
// the same residue r = oy % d.  Instead of the whole map for a 16-channel slab (64 bytes per pixel: every global access of
// dw3x3_map_lds_kernel is half a cache line at a 23 KB stride, and at 720x1280 x 64 frames the HBM sees 2.7 GB of them at
// 4.15 TB/s) a workgroup stages the <= ceil(H / d) rows of ONE class for a slab of 4 * rc_q[group] channels -- the fewer rows a
// class has, the wider the slab (launch_rowclass): 64 channels for the four rows of a d = 6 class of the 23x40 map, 128 for the
// one or two rows at d = 12 / 18; 256-512 contiguous bytes per pixel, <= 40 KB of LDS, three workgroups per CU.
// Blocks are numbered [dilation group][image][class][slab] with the slab fastest, and renumbered per XCD: the workgroups running
// side by side on an XCD read neighbouring pieces of the same pixels.  Each byte is still fetched once (PMC: 2.73 GB for 2.71).
// Item i of the LDS image = (class row j, column x, channel quad q), lane-linear (LDS-DMA); a thread keeps one channel quad.
// 652 -> 577 us in the 720x1280 plan (0.520 -> 0.588 of 8 TB/s); profiles/r5_experiments.md has the versions.
template <int NT>
__global__ __launch_bounds__(NT) void dw3x3_rowclass_kernel(const DwK p) {
    extern __shared__ __attribute__((aligned(16))) float smap[];
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    int vb = xcd_virtual_block(blockIdx.x, gridDim.x);
    int g = 0;
    while (g < 3 && vb >= p.rc_first[g + 1]) ++g;
    vb -= p.rc_first[g];
    const int Q = p.rc_q[g], CB = Q * 4;             // (NT % Q == 0: a thread keeps one channel quad)
    const int C = p.C4 * 4;
    const int gc0 = p.dgc ? g * p.dgc : 0;
    const int gcw = p.dgc ? (C - gc0 < p.dgc ? C - gc0 : p.dgc) : C;          // channels of this group
    const int d = p.dgc ? p.dils[g] : p.dil;
    const int slabs = (gcw + CB - 1) / CB, classes = d < p.H ? d : p.H;
    const int slab = vb % slabs; vb /= slabs;
    const int r = vb % classes;
    const int n = vb / classes;
    const int k = (p.H - 1 - r) / d + 1;             // rows r, r + d, ... of this class
    const int q = threadIdx.x % Q;
    const int cl = slab * CB + q * 4;                // channel inside the group
    const bool cok = cl < gcw;
    const int c = gc0 + (cok ? cl : 0);
    const int items = k * p.W * Q;
    const float* inb = p.in + (size_t)n * p.H * p.W * p.ldi + c;
    const f32x4* sm = reinterpret_cast<const f32x4*>(smap);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i0 = 0; i0 < items; i0 += NT) {         // (the LDS image is padded to whole NT-item rounds: launch_rowclass)
        const int i = i0 + threadIdx.x;
        const int px = i / Q, j = px / p.W, x = px - j * p.W;
        const float* src = (cok && i < items) ? inb + ((size_t)(r + j * d) * p.W + x) * p.ldi : g_dw_zero;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(smap + (size_t)(i0 + wave * 64) * 4), 16, 0, 0);
    }
    f32x4 wt[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wt[t] = ld4(p.w9c + (size_t)t * C + c);
    const f32x4 sc = ld4(p.scale + c), bi = ld4(p.bias + c);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (!cok) return;
    float* outb = p.out + (size_t)n * p.H * p.W * p.ldo + c;
    for (int i = threadIdx.x; i < items; i += NT) {
        const int px = i / Q, j = px / p.W, x = px - j * p.W;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int jj = j + ky - 1;
            if (jj < 0 || jj >= k) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = x + (kx - 1) * d;
                if (ix < 0 || ix >= p.W) continue;
                acc += sm[(jj * p.W + ix) * Q + q] * wt[ky * 3 + kx];
            }
        }
        f32x4 v = acc * sc + bi;
        if (p.act == UAVSAL_ACT_RELU6) {
            v.x = fminf(fmaxf(v.x, 0.f), 6.f); v.y = fminf(fmaxf(v.y, 0.f), 6.f);
            v.z = fminf(fmaxf(v.z, 0.f), 6.f); v.w = fminf(fmaxf(v.w, 0.f), 6.f);
        }
        *reinterpret_cast<f32x4*>(outb + ((size_t)(r + j * d) * p.W + x) * p.ldo) = v;
    }
}

// quads per slab of a group: the widest of 64 / 32 / 16 whose largest class stays within `budget` items (0: none does);
// the group's channels must be whole slabs of it
static inline int rowclass_quads(const DwK& k, int g, long long budget) {
    const int d = k.dgc ? k.dils[g] : k.dil;
    const int C = k.C4 * 4, gcw = k.dgc ? (C - g * k.dgc < k.dgc ? C - g * k.dgc : k.dgc) : C;
    const long long rows = (k.H - 1) / d + 1;
    for (int q = 64; q >= 8; q >>= 1)
        if (rows * k.W * q <= budget && gcw % (4 * q) == 0) return q;
    return 0;
}

static const long long ROWCLASS_ITEMS = [] { const char* e = getenv("UAVSAL_DW_ROWCLASS_ITEMS"); return e ? atoll(e) : 2560LL; }();       // 40 KB of LDS per workgroup: three per CU

static inline bool rowclass_fits(const DwK& k) {
    const int groups = k.dgc ? (k.C4 * 4 + k.dgc - 1) / k.dgc : 1;
    for (int g = 0; g < groups && g < 4; ++g)
        if (!rowclass_quads(k, g, ROWCLASS_ITEMS)) return false;
    return true;
}

template <int NT>
int launch_rowclass(DwK k, hipStream_t s) {
    const int C = k.C4 * 4;
    const int groups = k.dgc ? (C + k.dgc - 1) / k.dgc : 1;
    long long first = 0, most = 0;
    for (int g = 0; g < 4; ++g) {
        k.rc_first[g] = (int)first;
        k.rc_q[g] = 16;
        if (g < groups) {
            const int d = k.dgc ? k.dils[g] : k.dil;
            const int gcw = k.dgc ? (C - g * k.dgc < k.dgc ? C - g * k.dgc : k.dgc) : C;
            const int q = rowclass_quads(k, g, ROWCLASS_ITEMS);
            if (!q) return UAVSAL_ESHAPE;
            k.rc_q[g] = q;
            const long long items = (long long)((k.H - 1) / d + 1) * k.W * q;
            if (items > most) most = items;
            first += (long long)k.n_img * (gcw / (4 * q)) * (d < k.H ? d : k.H);
        }
        if (first > 0x7fffffffLL) return UAVSAL_ESHAPE;
    }
    k.rc_first[4] = (int)first;
    const size_t smem = (size_t)((most + NT - 1) / NT * NT) * 16;
    UAVSAL_LDS_OPTIN((&dw3x3_rowclass_kernel<NT>), smem);
    hipLaunchKernelGGL((dw3x3_rowclass_kernel<NT>), dim3((unsigned)first), dim3(NT), smem, s, k);
    return uavsal_launch_status();
}

// LDS image of a slab: [pixel][CB / 4] float4 items, padded to whole rounds of 256 items (one LDS-DMA request per wave and round)
static inline size_t map_lds_bytes(long long px, int cb) { return (size_t)((px * (cb / 4) + 1023) / 1024 * 1024) * 16; }

// threads per workgroup: 512 where a slab is many rounds of 256 items (the 23x40 map of 720x1280 inputs: 14.4 rounds; two
// workgroups per CU by LDS either way, twice the waves to cover the store phase: 0.47 -> 0.52 of 8 TB/s at 64 frames)
static inline int map_lds_threads(const DwK& k, int cb) {
    static const int nt_forced = [] { const char* e = getenv("UAVSAL_DW_MAP_NT"); return e ? atoi(e) : 0; }();
    if (nt_forced == 256 || nt_forced == 512 || nt_forced == 1024) return nt_forced;
    return (long long)k.H * k.W * (cb / 4) >= 2048 ? 512 : 256;
}

template <int CB>
int launch_map_lds(DwK k, hipStream_t s) {
    const int slabs = (k.C4 * 4 + CB - 1) / CB;
    const size_t smem = map_lds_bytes((long long)k.H * k.W, CB);
    // 512-thread workgroups where a slab is many rounds of 256 items (the 23x40 map of 720x1280 inputs: 14.4): two workgroups per
    // CU by LDS either way, twice the waves to cover the store phase
    const int nt = map_lds_threads(k, CB);
    if (nt == 1024) hipLaunchKernelGGL((dw3x3_map_lds_kernel<CB, 1024>), dim3((unsigned)(k.n_img * slabs)), dim3(1024), smem, s, k);
    else if (nt == 512) hipLaunchKernelGGL((dw3x3_map_lds_kernel<CB, 512>), dim3((unsigned)(k.n_img * slabs)), dim3(512), smem, s, k);
    else hipLaunchKernelGGL((dw3x3_map_lds_kernel<CB, 256>), dim3((unsigned)(k.n_img * slabs)), dim3(256), smem, s, k);
    return uavsal_launch_status();
}

// channel slab width for the whole-map kernel, 0 = the map does not fit 64 KB of LDS at 16 channels
static inline int map_lds_slab(const DwK& k) {
    const long long px = (long long)k.H * k.W;
    if (map_lds_bytes(px, 16) > 65536) return 0;
    static const int forced = [] { const char* e = getenv("UAVSAL_DW_MAP_CB"); return e ? atoi(e) : 0; }();
    if ((forced == 16 || forced == 32 || forced == 64) && map_lds_bytes(px, forced) <= 65536) return forced;
    // 32 channels (30 KB of LDS at 12x20: five workgroups per CU cover each other's load / compute phases) where that still
    // gives every CU a workgroup, else 16.  (64-channel slabs, two workgroups per CU: 78.7 vs 51.9 us at 64 x 12x20 x 1920
    // and 15.1 vs 11.0 us at 8 frames -- round 4)
    const int cands[2] = {32, 16};
    for (int i = 0; i < 2; ++i) {
        const int cb = cands[i];
        if (map_lds_bytes(px, cb) <= 65536 && (long long)k.n_img * ((k.C4 * 4 + cb - 1) / cb) >= 256) return cb;
    }
    return 16;
}

// which kernel a descriptor gets: 1 = dw3x3_kernel<1,4,4>, 2 = <1,2,2>, 3 = <2,2,2>, 4 = dw3x3_dilated_kernel,
// 16 / 32 / 64 = dw3x3_map_lds_kernel<CB> (uavsal_dw_variant adds 512 / 1024 for the instances with that many threads)
static int dw_variant(const DwK& k, int stride, int dilation) {
    if (k.dgc) {              // several dilated branches in one launch: the whole-map kernel (each byte fetched once whatever the dilation)
        const int cb = map_lds_slab(k);
        // a map too big for 32-channel slabs (23x40, the 1/32 level of 720x1280 inputs) would be read and written in half cache
        // lines: by row class instead (measured at 64 frames x 5760 channels, in the plan: 577 vs 652 us, 0.588 vs 0.520 of 8 TB/s;
        // on the 12x20 map of 360x640 inputs the whole-map kernel wins, 20 vs 26 us at 8 frames and 142 vs 181 at 64).
        // UAVSAL_DW_ROWCLASS=0 / 1: never / wherever it fits
        static const int rowclass = [] { const char* e = getenv("UAVSAL_DW_ROWCLASS"); return e ? atoi(e) : -1; }();
        const bool big_map = map_lds_bytes((long long)k.H * k.W, 32) > 65536;
        if (rowclass != 0 && (rowclass > 0 || big_map) && rowclass_fits(k)) return 2048;
        return (cb && k.dgc % cb == 0) ? cb : 4;
    }
    // Measured at 8 x 12x20 x 1920 (profiles/r2_dw_small_maps.md): the whole-map LDS kernel wins only while most
    // taps fall inside the map (d=6: 10.9 vs 15.1 us); d=12/18 (centre row only) and dilation 1 (2x2 patches
    // already fetch every input 2.25x, from L1/L2) are faster on the direct kernels (8.7 / 7.7 us vs 9.3 / 9.4;
    // 7.3 vs 18 us at 23x40).
    if (stride == 1 && dilation != 1 && 2 * dilation <= (k.H < k.W ? k.H : k.W)) {
        const int cb = map_lds_slab(k);
        if (cb) return cb;
    }
    if (dilation != 1) return 4;
    if (stride == 1) {
        // 4x4 patches need ~2x fewer loads per output, but on the 12x20 / 23x40 maps they leave most of the
        // chip without a workgroup: below 2 workgroups per CU's worth of patches use 2x2 ones (4x the threads)
        const long long wg44 = ((long long)k.n_img * ((k.Ho + 3) / 4) * ((k.Wo + 3) / 4) * k.C4 + 255) / 256;
        return wg44 < 512 ? 2 : 1;
    }
    return 3;
}

template <int S, int TY, int TX>
int launch_dw(DwK k, hipStream_t s) {
    k.tiles_x = (k.Wo + TX - 1) / TX;
    k.tiles_y = (k.Ho + TY - 1) / TY;
    k.total = (long long)k.n_img * k.tiles_y * k.tiles_x * k.C4;
    const long long nblk = (k.total + 255) / 256;
    if (nblk > 0x7fffffffLL) return UAVSAL_ESHAPE;
    k.nblk = (int)nblk;
    hipLaunchKernelGGL((dw3x3_kernel<S, TY, TX>), dim3(k.nblk), dim3(256), 0, s, k);
    return uavsal_launch_status();
}

// ---- depthwise 3x3 (stride 1) + BN + ReLU6 -> 1x1 projection to ONE channel + BN + activation, one launch (uavsal_dw3x3_dot) ----
// The decoder's last block (conv_out_st = dwBlock(256 -> 1), reference model.py:333-334, 372-373) ends in a 1536 -> 1 projection:
// a dot product per pixel.  As a GEMM (dwproj_kernel with 31 of its 32 output columns zero + a K-split reduce launch) it moved
// the 177 MB of the expanded tensor at 2.7 TB/s; here it is the depthwise kernel above -- 4 channels per lane, a wave = 256
// consecutive channels of a pixel per load, a 4 x 4 output patch per thread -- whose epilogue multiplies by the lane's four
// projection weights instead of storing: the sixteen partial dot products are summed over the wave by a shuffle butterfly and
// over the workgroup's waves (one workgroup = ALL channels of one patch, blockDim = C / 4) through LDS, in a fixed order.
struct DwDotK {
    const float* in; const float* w9c; const float* scale; const float* bias; const float* w2;
    const float* scale2; const float* bias2; float* out;
    int ldi, ldo, H, W, C4, act, tiles_x, tiles_y, nblk;
};

// (230 VGPRs: one 6-wave workgroup per CU at C = 1536.  Capped at 168 / 128 for two or three -- __launch_bounds__(512, 3 | 4), 4x4 /
// 2x8 / 2x4 patches -- hipcc spills 29-77 registers and the launch takes 95-123 us instead of 52: profiles/r4_dw_dot.md)
template <int TY, int TX>
__global__ __launch_bounds__(512, 2) void dw3x3_dot_kernel(const DwDotK p) {
    __shared__ float red[8][TY * TX];
    int t = xcd_virtual_block(blockIdx.x, p.nblk);
    const int tx = t % p.tiles_x; t /= p.tiles_x;
    const int ty = t % p.tiles_y;
    const int n = t / p.tiles_y;
    const int c = threadIdx.x * 4;
    f32x4 wt[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wt[k] = ld4(p.w9c + (size_t)k * (p.C4 * 4) + c);
    const f32x4 sc = ld4(p.scale + c), bi = ld4(p.bias + c), w2 = ld4(p.w2 + c);
    constexpr int IW = TX + 2, IH = TY + 2;
    const int oy0 = ty * TY, ox0 = tx * TX;
    const float* inb = p.in + (size_t)n * p.H * p.W * p.ldi + c;
    f32x4 acc[TY][TX];
#pragma unroll
    for (int a = 0; a < TY; ++a)
#pragma unroll
        for (int b = 0; b < TX; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < IH; ++r) {
        const int iy = oy0 - 1 + r;
        const bool rok = iy >= 0 && iy < p.H;
        f32x4 row[IW];
#pragma unroll
        for (int q = 0; q < IW; ++q) {
            const int ix = ox0 - 1 + q;
            const bool ok = rok && ix >= 0 && ix < p.W;
            row[q] = ok ? ld4(inb + ((size_t)iy * p.W + ix) * p.ldi) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int a = 0; a < TY; ++a) {
            const int ky = r - a;
            if (ky < 0 || ky > 2) continue;
#pragma unroll
            for (int b = 0; b < TX; ++b)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) acc[a][b] += row[b + kx] * wt[ky * 3 + kx];
        }
    }
    float part[TY * TX];
#pragma unroll
    for (int a = 0; a < TY; ++a)
#pragma unroll
        for (int b = 0; b < TX; ++b) {
            f32x4 v = acc[a][b] * sc + bi;
            v.x = fminf(fmaxf(v.x, 0.f), 6.f); v.y = fminf(fmaxf(v.y, 0.f), 6.f);
            v.z = fminf(fmaxf(v.z, 0.f), 6.f); v.w = fminf(fmaxf(v.w, 0.f), 6.f);
            part[a * TX + b] = (v.x * w2.x + v.y * w2.y) + (v.z * w2.z + v.w * w2.w);
        }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
        for (int i = 0; i < TY * TX; ++i) part[i] += __shfl_xor(part[i], off, 64);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int i = 0; i < TY * TX; ++i) red[wave][i] = part[i];
    }
    __syncthreads();
    if (threadIdx.x < TY * TX) {
        const int a = threadIdx.x / TX, b = threadIdx.x - a * TX;
        const int oy = oy0 + a, ox = ox0 + b;
        if (oy < p.H && ox < p.W) {
            float sum = 0.f;
            const int nw = (int)(blockDim.x >> 6);
            for (int w = 0; w < nw; ++w) sum += red[w][threadIdx.x];
            float z = sum * p.scale2[0] + p.bias2[0];
            if (p.act == UAVSAL_ACT_SIGMOID) z = 1.f / (1.f + expf(-z));
            else if (p.act == UAVSAL_ACT_RELU6) z = fminf(fmaxf(z, 0.f), 6.f);
            p.out[(((size_t)n * p.H + oy) * p.W + ox) * p.ldo] = z;
        }
    }
}


}  // namespace

extern "C" int uavsal_dw3x3(const uavsal_dw_desc* d, uavsal_stream_t stream) {
    if (!d || !d->in || !d->w9c || !d->scale || !d->bias) return UAVSAL_EINVAL;
    if (!d->out && !d->out_split) return UAVSAL_EINVAL;
    if (d->n_img <= 0 || d->H <= 0 || d->W <= 0 || d->C <= 0) return UAVSAL_EINVAL;
    if ((d->C & 3) || (d->ldi & 3) || d->ldi < d->C) return UAVSAL_EALIGN;
    if (d->out && ((d->ldo & 3) || d->ldo < d->C || !uavsal_aligned16(d->out))) return UAVSAL_EALIGN;
    if (d->out_split && ((d->ldos & 63) || d->ldos < 2 * d->C || (d->C & 31) || ((uintptr_t)d->out_split & 127)))
        return UAVSAL_EALIGN;
    if (d->out_split && d->dilation != 1) return UAVSAL_ESHAPE;
    if (!uavsal_aligned16(d->in) || !uavsal_aligned16(d->w9c) ||
        !uavsal_aligned16(d->scale) || !uavsal_aligned16(d->bias)) return UAVSAL_EALIGN;
    if (d->stride != 1 && d->stride != 2) return UAVSAL_ESHAPE;
    if (d->dilation < 1 || (d->stride == 2 && d->dilation != 1)) return UAVSAL_ESHAPE;
    DwK k;
    k.in = d->in; k.w9c = d->w9c; k.scale = d->scale; k.bias = d->bias; k.out = d->out;
    k.out_split = (_Float16*)d->out_split; k.ldos = d->ldos;
    k.ldi = d->ldi; k.ldo = d->ldo; k.H = d->H; k.W = d->W;
    k.Ho = (d->H - 1) / d->stride + 1; k.Wo = (d->W - 1) / d->stride + 1;
    k.C4 = d->C / 4; k.dil = d->dilation; k.act = d->act; k.n_img = d->n_img;
    k.tiles_x = k.tiles_y = 0; k.total = 0; k.nblk = 0;
    k.dgc = 0; k.dils[0] = k.dils[1] = k.dils[2] = k.dils[3] = d->dilation;
    if (d->dil_group_c) {
        if (d->dil_group_c < 0 || (d->dil_group_c & 3) || d->stride != 1 || d->out_split || d->C > 4 * d->dil_group_c) return UAVSAL_ESHAPE;
        for (int i = 0; i < 4; ++i) {
            if ((long long)i * d->dil_group_c < d->C && d->dil_groups[i] < 1) return UAVSAL_ESHAPE;
            k.dils[i] = d->dil_groups[i] > 0 ? d->dil_groups[i] : 1;
        }
        k.dgc = d->dil_group_c;
    }
    hipStream_t s = (hipStream_t)stream;
    const int v = dw_variant(k, d->stride, d->dilation);
    switch (v) {
        case 2048: { static const int nt = [] { const char* e = getenv("UAVSAL_DW_ROWCLASS_NT"); return e ? atoi(e) : 256; }();
                     return nt == 256 ? launch_rowclass<256>(k, s) : launch_rowclass<512>(k, s); }
        case 64: return launch_map_lds<64>(k, s);
        case 32: return launch_map_lds<32>(k, s);
        case 16: return launch_map_lds<16>(k, s);
        case 4: {
            k.total = (long long)k.n_img * k.Ho * k.Wo * k.C4;
            const long long nblk = (k.total + 255) / 256;
            if (nblk > 0x7fffffffLL) return UAVSAL_ESHAPE;
            k.nblk = (int)nblk;
            hipLaunchKernelGGL(dw3x3_dilated_kernel, dim3(k.nblk), dim3(256), 0, s, k);
            return uavsal_launch_status();
        }
        case 1: return launch_dw<1, 4, 4>(k, s);
        case 2: return launch_dw<1, 2, 2>(k, s);
        default: return launch_dw<2, 2, 2>(k, s);
    }
}

extern "C" int uavsal_dw_variant(const uavsal_dw_desc* d) {
    if (!d || d->n_img <= 0 || d->H <= 0 || d->W <= 0 || d->C <= 0 || (d->C & 3)) return UAVSAL_EINVAL;
    if ((d->stride != 1 && d->stride != 2) || d->dilation < 1 || (d->stride == 2 && d->dilation != 1)) return UAVSAL_ESHAPE;
    DwK k;
    k.H = d->H; k.W = d->W; k.Ho = (d->H - 1) / d->stride + 1; k.Wo = (d->W - 1) / d->stride + 1;
    k.C4 = d->C / 4; k.n_img = d->n_img; k.dgc = d->dil_group_c > 0 ? d->dil_group_c : 0;
    k.dil = d->dilation;
    for (int i = 0; i < 4; ++i) k.dils[i] = d->dil_groups[i] > 0 ? d->dil_groups[i] : 1;
    const int v = dw_variant(k, d->stride, d->dilation);
    if (v >= 2048) return v;
    const int nt = v >= 16 ? map_lds_threads(k, v) : 256;
    return nt == 256 ? v : v + nt;
}

extern "C" int uavsal_dw3x3_dot(const uavsal_dw_dot_desc* d, uavsal_stream_t stream) {
    if (!d || !d->in || !d->w9c || !d->scale || !d->bias || !d->w2 || !d->scale2 || !d->bias2 || !d->out) return UAVSAL_EINVAL;
    if (d->n_img <= 0 || d->H <= 0 || d->W <= 0 || d->C <= 0 || d->ldo < 1) return UAVSAL_EINVAL;
    if ((d->C & 255) || d->C > 2048) return UAVSAL_ESHAPE;          // one workgroup = all channels of a patch: whole waves, <= 512 lanes
    if ((d->ldi & 3) || d->ldi < d->C) return UAVSAL_EALIGN;
    if (!uavsal_aligned16(d->in) || !uavsal_aligned16(d->w9c) || !uavsal_aligned16(d->scale) || !uavsal_aligned16(d->bias) ||
        !uavsal_aligned16(d->w2)) return UAVSAL_EALIGN;
    if (d->act != UAVSAL_ACT_NONE && d->act != UAVSAL_ACT_RELU6 && d->act != UAVSAL_ACT_SIGMOID) return UAVSAL_ESHAPE;
    DwDotK k;
    k.in = d->in; k.w9c = d->w9c; k.scale = d->scale; k.bias = d->bias; k.w2 = d->w2; k.scale2 = d->scale2; k.bias2 = d->bias2;
    k.out = d->out; k.ldi = d->ldi; k.ldo = d->ldo; k.H = d->H; k.W = d->W; k.C4 = d->C / 4; k.act = d->act;
#ifndef UAVSAL_DWDOT_TY
#define UAVSAL_DWDOT_TY 4
#endif
#ifndef UAVSAL_DWDOT_TX
#define UAVSAL_DWDOT_TX 4
#endif
    k.tiles_x = (d->W + UAVSAL_DWDOT_TX - 1) / UAVSAL_DWDOT_TX; k.tiles_y = (d->H + UAVSAL_DWDOT_TY - 1) / UAVSAL_DWDOT_TY;
    const long long nblk = (long long)d->n_img * k.tiles_y * k.tiles_x;
    if (nblk > 0x7fffffffLL) return UAVSAL_ESHAPE;
    k.nblk = (int)nblk;
    hipLaunchKernelGGL((dw3x3_dot_kernel<UAVSAL_DWDOT_TY, UAVSAL_DWDOT_TX>), dim3(k.nblk), dim3(k.C4), 0, (hipStream_t)stream, k);
    return uavsal_launch_status();
}

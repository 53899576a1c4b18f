// Fused inverted-residual block for the MID-channel blocks on small maps (round 4):
//   out = bn3(pw_linear( relu6(bn2(dw3x3( relu6(bn1(pw_expand(x))) ))) )) [+ x]        stride 1
// torchvision InvertedResidual / dwBlock (reference model.py:74-103) for Cin 64 / 96, hidden 384 / 576 -- MobileNetV2
// features[8..13] on the 1/16-scale map (23x40 at 360x640), which the engine used to run as three launches of 7-18 us each
// on the per-launch floor (ten blocks = 30 launches = 0.37 ms of a 4.4 ms step).  `fused_ir_kernel` (fused_ir.hip) is built
// for the tiny-channel head of the backbone -- register-resident weight fragments, 16 hidden channels per chunk: 24-36
// chunks in sequence here -- and loses to the three launches on these shapes (profiles/r3_fused_ir.md).  This kernel is the
// GEMM-shaped form of the same fusion:
//   * one 256-thread workgroup owns a 4 x 8 patch of output pixels (23x40 x 8 frames: 240 workgroups, one per CU);
//   * x of the 6 x 10 halo (64 rows, 4 of them padding) is staged ONCE in LDS by LDS-DMA and its MFMA fragments are kept in
//     registers for every hidden chunk (the depthwise output D later reuses that LDS space);
//   * the hidden channels are walked 64 at a time; the chunk's weights -- W1[64 hidden][Cin], W2[Cout][64 hidden], both in
//     the conv weight's natural layout -- and its parameter vectors (9 taps + BN2, BN1) arrive by LDS-DMA a chunk ahead, two
//     buffers each, from running per-lane source pointers; every wave issues the same 9-13 requests per chunk, one per slice
//     of the iteration (below);
//   * expand:  E^T[hidden][pixel] = W1c . x^T on v_mfma_f32_32x32x2_f32 (exact fp32), four 32 x 32 tiles = one per wave;
//     the accumulator lane holds 4 x 4 consecutive hidden channels of one pixel: relu6(bn1), EXACT ZERO outside the image
//     (the depthwise conv pads E, not x), four ds_write_b128 into E[pixel][64];
//   * depthwise: a thread owns (a pair of horizontally adjacent pixels, 4 channels): its 3 x 4 halo pixels are read once for
//     both (12 ds_read_b128), taps / BN2 from LDS, relu6(bn2) -> D[pixel][64];
//   * project: OUT^T[cout][pixel] += W2c . D^T, every wave takes a quarter of the chunk's K for all Cout tiles; the four
//     partial accumulators are summed once at the end through LDS in wave order (deterministic), then bn3 / residual / store;
//   * ONE barrier per chunk: iteration c runs expand(c + 1), depthwise(c) and project(c - 1) (E and D double-buffered), and
//     because a one-wave-per-SIMD kernel gets no overlap from the hardware, the iteration is hand-cut into slices
//     [first MFMA of a group][one halo pixel's FMAs] | [the slice's LDS-DMA request] | [LDS reads of the NEXT slice] |
//     [the group's other three MFMAs], pinned with sched_barrier / asm volatile.  profiles/r4_fused_mid.md has the versions,
//     the in-kernel stamps and what bounds it now (the weight stream: 35 KB per chunk and workgroup by LDS-DMA).
// LDS rows are 256 B (or 384 B) with the 16-byte chunk index XOR-swizzled by the row (conflict-free ds_read_b128 fragment
// reads); the swizzle is applied on the SOURCE address of the DMA requests (their LDS image is lane-linear).
// MFMA lane maps (cdna_hip_programming.md; the same as conv_gemm_k32.hip): A[m = l & 31][k from l >> 5], B[k][n = l & 31],
// D[(g & 3) + 8 (g >> 2) + 4 (l >> 5)][l & 31].
#include "common.h"
#include <type_traits>

namespace {

struct MidK {
    const float* in; const float* w1; const float* s1; const float* b1;
    const float* wd; const float* sd; const float* bd;
    const float* w2; const float* s2; const float* b2;
    const float* res; float* out;
    int ldi, ldr, ldo, H, W, tiles_x, tiles_y;
};

__device__ __attribute__((aligned(16))) float g_mid_zero[4];
#ifndef UAVSAL_MID_PROBE
#define UAVSAL_MID_PROBE 0     /* timing experiments only (wrong results): 1 no MFMAs, 2 no depthwise (halo reads + FMAs), 4 no vector-memory
                                  instructions in the iterations, 8 no fragment reads */
#endif
#ifdef UAVSAL_MID_STAMPS      // diagnostic build only (tools/mid_probe.py): s_memtime at the phase boundaries of workgroup 0, wave 0
__device__ unsigned long long g_mid_stamps[32];
#define MID_STAMP(i) { if (blockIdx.x == 0 && tid == 0) g_mid_stamps[i] = __builtin_amdgcn_s_memtime(); }
#else
#define MID_STAMP(i)
#endif

// physical 16-byte chunk of logical chunk c in row `row` of an LDS panel with NC chunks per row
template <int NC>
__device__ __forceinline__ int mid_swz(int row, int c) {
    if (NC % 16 == 0) return c ^ (row & 15);                   // 256-byte rows: every row starts on bank 0
    return (c & ~7) | ((c & 7) ^ ((row >> 1) & 7));             // 384-byte rows: row parity selects the bank half
}

__device__ __forceinline__ f32x4 mid_relu6(f32x4 v) {
    return (f32x4){__builtin_amdgcn_fmed3f(v.x, 0.f, 6.f), __builtin_amdgcn_fmed3f(v.y, 0.f, 6.f), __builtin_amdgcn_fmed3f(v.z, 0.f, 6.f),
                   __builtin_amdgcn_fmed3f(v.w, 0.f, 6.f)};
}

template <int CIN, int HID, int COUT>
struct MidCfg {
    static constexpr int PH = 4, PW = 8, HH = PH + 2, HWD = PW + 2, NHALO = HH * HWD;       // 6 x 10 = 60 halo pixels
    static constexpr int HC = 64, NCH = HID / HC, CT = COUT / 32;
    static constexpr int XC = CIN / 4;                          // 16-byte chunks per x / W1 row
    static constexpr int X_F = 64 * CIN, W1_F = 64 * CIN, W2_F = COUT * HC, E_F = 64 * HC, D_F = 32 * HC;
    // two buffers each of W1 / W2 / E / D; x is only needed until its fragments are in registers: D reuses its space
    static constexpr int XD_F = X_F > 2 * D_F ? X_F : 2 * D_F;
    // PT: the depthwise taps + BN2 of a chunk (11 rows of 64 floats, three 1 KB requests); PB: BN1 (2 rows, one request)
    static constexpr int PT_F = 768, PB_F = 256;
    static constexpr int OFF_W1 = 0, OFF_W2 = OFF_W1 + 2 * W1_F, OFF_E = OFF_W2 + 2 * W2_F, OFF_PT = OFF_E + 2 * E_F,
                         OFF_PB = OFF_PT + 2 * PT_F, OFF_X = OFF_PB + 2 * PB_F, OFF_D = OFF_X, TOTAL_F = OFF_X + XD_F;
    static constexpr int RED_F = 4 * CT * 1024;                 // the four waves' partial projections (reuse W1 / W2)
    static constexpr size_t SMEM = (size_t)TOTAL_F * 4;
    static_assert(CIN % 32 == 0 && HID % 64 == 0 && COUT % 32 == 0 && NHALO <= 64, "blocking");
    static_assert(RED_F <= 2 * W1_F + 2 * W2_F, "the partial sums fit in the weight buffers");
    static_assert(SMEM <= 160 * 1024, "LDS");
};

template <int CIN, int HID, int COUT>
__global__ __launch_bounds__(256, 1) void fused_mid_kernel(const MidK p) {
    using Cfg = MidCfg<CIN, HID, COUT>;
    constexpr int PH = Cfg::PH, PW = Cfg::PW, HWD = Cfg::HWD, NHALO = Cfg::NHALO, HC = Cfg::HC, NCH = Cfg::NCH, CT = Cfg::CT, XC = Cfg::XC;
    constexpr int KG = CIN / 8;                  // groups of 8 input channels = 4 MFMAs each
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const Xs = lds + Cfg::OFF_X;
    float* const W1s = lds + Cfg::OFF_W1;
    float* const W2s = lds + Cfg::OFF_W2;
    float* const Es = lds + Cfg::OFF_E;
    float* const Ds = lds + Cfg::OFF_D;
    float* const PTs = lds + Cfg::OFF_PT;
    float* const PBs = lds + Cfg::OFF_PB;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    int t = blockIdx.x;
    const int tx = t % p.tiles_x; t /= p.tiles_x;
    const int ty = t % p.tiles_y;
    const int n = t / p.tiles_y;
    const int oy0 = ty * PH, ox0 = tx * PW;
    const float* inb = p.in + (size_t)n * p.H * p.W * p.ldi;

    // halo row r (0..63) -> is it a pixel of the image
    auto halo_ok = [&](int r, int& gy, int& gx) -> bool {
        const int hy = r / HWD, hx = r - hy * HWD;
        gy = oy0 - 1 + hy; gx = ox0 - 1 + hx;
        return r < NHALO && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
    };

    // ---- LDS-DMA requests: a wave request = 1 KB, lane-linear in LDS; lane i lands at byte q * 1024 + i * 16 of the panel.
    // Request i of this wave is q = wave + 4 i; its source is a running per-lane pointer (64 W1 rows / 64 W2 columns further
    // per chunk): no address arithmetic beyond one 64-bit add per request and chunk.
    constexpr int XREQ = 64 * CIN * 4 / 1024 / 4, W2REQ = COUT * HC * 4 / 1024 / 4;
    const float* w1_src[XREQ];
    const float* w2_src[W2REQ];
#pragma unroll
    for (int i = 0; i < XREQ; ++i) {
        const int off = (wave + 4 * i) * 1024 + lane * 16;
        const int row = off / (CIN * 4), pc = (off - row * (CIN * 4)) / 16;
        w1_src[i] = p.w1 + (size_t)row * CIN + mid_swz<XC>(row, pc) * 4;
        int gy, gx;
        const bool ok = halo_ok(row, gy, gx);
        const float* src = ok ? inb + ((size_t)gy * p.W + gx) * p.ldi + mid_swz<XC>(row, pc) * 4 : g_mid_zero;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Xs + (wave + 4 * i) * 256), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < W2REQ; ++i) {
        const int off = (wave + 4 * i) * 1024 + lane * 16;
        const int row = off / 256, pc = (off & 255) / 16;
        w2_src[i] = p.w2 + (size_t)row * HID + mid_swz<16>(row, pc) * 4;
    }
    // the per-chunk parameter vectors travel the same way (a global load per lane and vector was 19 vector-memory instructions
    // per wave and chunk, and the vector-memory issue of the CU -- not the matrix pipe -- set the length of an iteration:
    // profiles/r4_fused_mid.md).  Waves 0-2 fetch a third each of PT(ch) = [9 taps | BN2 scale | BN2 bias][64], wave 3 fetches
    // PB(ch) = [BN1 scale | BN1 bias][64]; lanes past the rows fetch the zero page.
    const float* par_src;
    {
        const int row = (wave < 3 ? wave * 4 : 0) + (lane >> 4), cc = (lane & 15) * 4;
        if (wave < 3) par_src = row < 9 ? p.wd + (size_t)row * HID + cc : (row == 9 ? p.sd + cc : (row == 10 ? p.bd + cc : nullptr));
        else par_src = row == 0 ? p.s1 + cc : (row == 1 ? p.b1 + cc : nullptr);
    }
    const bool par_ok = par_src != nullptr;
    if (!par_ok) par_src = g_mid_zero;
    auto dma_par = [&](int ch) {                     // waves 0-2: their part of PT(ch) -> buffer ch & 1;  wave 3: PB(ch) -> buffer ch & 1
        float* dst = wave < 3 ? PTs + (ch & 1) * Cfg::PT_F + wave * 256 : PBs + (ch & 1) * Cfg::PB_F;
        __builtin_amdgcn_global_load_lds((gptr_t)(par_ok ? par_src + ch * HC : par_src), (lptr_t)dst, 16, 0, 0);
    };
    auto dma_w1_one = [&](int ch, int i) {           // request i of the NEXT W1 chunk in sequence (chunks are requested in order) -> buffer ch & 1
        __builtin_amdgcn_global_load_lds((gptr_t)w1_src[i], (lptr_t)(W1s + (ch & 1) * Cfg::W1_F + (wave + 4 * i) * 256), 16, 0, 0);
        w1_src[i] += HC * CIN;
    };
    auto dma_w2_one = [&](int ch, int i) {           // request i of the next W2 chunk in sequence -> buffer ch & 1
        __builtin_amdgcn_global_load_lds((gptr_t)w2_src[i], (lptr_t)(W2s + (ch & 1) * Cfg::W2_F + (wave + 4 * i) * 256), 16, 0, 0);
        w2_src[i] += HC;
    };
    auto dma_w1 = [&](int ch) {
#pragma unroll
        for (int i = 0; i < XREQ; ++i) dma_w1_one(ch, i);
    };

    // this wave's expand tile: hidden rows ht * 32 .., halo pixels nt * 32 ..; its x fragments stay in registers
    const int ht = wave & 1, nt = wave >> 1;
    f32x4 xf[KG];
    int gy_, gx_;
    const bool px_ok = halo_ok(nt * 32 + lr, gy_, gx_);          // E of this lane's pixel is zero outside the image

    f32x16 acc_o[CT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int g = 0; g < 16; ++g) acc_o[ct][g] = 0.f;

    // depthwise item of this thread: the pixel pair (oy, ox), (oy, ox + 1) of the 4 x 8 patch x channel quad tid >> 4: its 3 x 4
    // halo pixels are read once for both (12 ds_read_b128 instead of 18); the taps + BN of the quad live in registers, loaded
    // from global memory a chunk ahead (the LDS is the busiest unit of this kernel: 220 KB per chunk and CU before this)
    const int cq = tid >> 4;
    const int pp_y = (tid & 15) >> 2, pp_x = 2 * (tid & 3);
    const int hs00 = pp_y * HWD + pp_x;                          // halo row of the pair's top-left tap
    const int cq4 = cq * 4;

    // expand of chunk 0 (prologue): E^T tile = W1c (A, from LDS) . x^T (B, registers) -> relu6(bn1), zero outside the image -> E[0]
    auto expand = [&](int ch) {
        f32x4 s1v[4], b1v[4];                    // BN1 of this lane's 16 hidden channels: 4 x 4 consecutive ones
        const float* pb = PBs + (ch & 1) * Cfg::PB_F + ht * 32 + 4 * lh;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            s1v[q] = *reinterpret_cast<const f32x4*>(pb + 8 * q);
            b1v[q] = *reinterpret_cast<const f32x4*>(pb + 64 + 8 * q);
        }
        f32x16 acc;
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[g] = 0.f;
        const float* w1b = W1s + (ch & 1) * Cfg::W1_F;
        const int row = ht * 32 + lr;
#pragma unroll
        for (int u = 0; u < KG; ++u) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(w1b + row * CIN + mid_swz<XC>(row, 2 * u + lh) * 4);
            const f32x4 bv = xf[u];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
        }
        float* eb = Es + (ch & 1) * Cfg::E_F;
        const int prow = nt * 32 + lr;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 e = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
            e = mid_relu6(e * s1v[q] + b1v[q]);
            if (!px_ok) e = (f32x4){0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(eb + prow * HC + mid_swz<16>(prow, ht * 8 + lh + 2 * q) * 4) = e;
        }
    };
    // project of the last chunk (after the loop): this wave's quarter of the chunk's K for every Cout tile, from D[ch & 1]
    auto project = [&](int ch) {
        const float* w2b = W2s + (ch & 1) * Cfg::W2_F;
        const float* db = Ds + (ch & 1) * Cfg::D_F;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int c = 4 * wave + 2 * u + lh;
            const f32x4 bv = *reinterpret_cast<const f32x4*>(db + lr * HC + mid_swz<16>(lr, c) * 4);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int row = ct * 32 + lr;
                const f32x4 av = *reinterpret_cast<const f32x4*>(w2b + row * HC + mid_swz<16>(row, c) * 4);
                acc_o[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc_o[ct], 0, 0, 0);
                acc_o[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc_o[ct], 0, 0, 0);
                acc_o[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc_o[ct], 0, 0, 0);
                acc_o[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc_o[ct], 0, 0, 0);
            }
        }
    };

    // ---- prologue: x (requested above), W1 of chunks 0 and 1, the depthwise taps of chunk 0; expand(0)
    MID_STAMP(0)
    dma_w1(0);
    if (NCH > 1) dma_w1(1);
    dma_par(0);                      // PT(0) / PB(0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    MID_STAMP(1)
    {
        const int row = nt * 32 + lr;
#pragma unroll
        for (int u = 0; u < KG; ++u) xf[u] = *reinterpret_cast<const f32x4*>(Xs + row * CIN + mid_swz<XC>(row, 2 * u + lh) * 4);
    }
    if (NCH > 1 && wave == 3) dma_par(1);      // PB(1): BN1 of expand(1), which runs in iteration 0 (PT(1) is requested there)
    expand(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                 // E[0] visible; every wave has its x fragments: D may now overwrite x's space
    MID_STAMP(2)

    // ---- software pipeline, ONE barrier per chunk: iteration c runs expand(c + 1), depthwise(c) and project(c - 1) -- three
    // chunks in flight, independent of each other inside the iteration -- and requests W1(c + 2) / W2(c) for the next
    // iteration (buffers last read an iteration ago).  One wave per SIMD, so the overlap has to come from the wave's own
    // instruction order: the iteration is cut into slices of [LDS reads for the NEXT slice: an MFMA group's fragments + one of
    // the depthwise's 12 halo pixels] [one group of 4 MFMAs: an expand K group, then the projection's] [the halo pixel's FMAs],
    // pinned with sched_barrier -- the depthwise's LDS / VALU work issues in the shadow of the MFMAs (hipcc left the three
    // stages back to back: profiles/r4_fused_mid.md).
    auto iteration = [&](auto has_exp, auto has_proj, int c) {
        constexpr bool EXP = decltype(has_exp)::value, PROJ = decltype(has_proj)::value;
        constexpr int GE = EXP ? KG : 0, GP = PROJ ? 2 * CT : 0, NG = GE + GP;
        constexpr int NS0 = NG > 13 ? NG : 13, NS1 = EXP && GE + 3 > NS0 ? GE + 3 : NS0, NS = XREQ + W2REQ + 1 > NS1 ? XREQ + W2REQ + 1 : NS1;
        // The iteration's LDS-DMA requests -- W1(c + 2), W2(c), PT(c + 1) (waves 0-2) / PB(c + 2) (wave 3) -- are dealt over
        // the slices below, one behind the first MFMA of a group: issued in one burst at the top they held the wave for
        // ~2000 cycles before its first MFMA (profiles/r4_fused_mid.md).
        const bool w1_more = c + 2 < NCH;
        const bool par_more = wave < 3 ? c + 1 < NCH : c + 2 < NCH;
        const int par_ch = wave < 3 ? c + 1 : c + 2;
        const float* pt = PTs + (c & 1) * Cfg::PT_F + cq4;          // taps / BN2 of chunk c, this thread's channel quad
        const float* pb = PBs + ((c + 1) & 1) * Cfg::PB_F + ht * 32 + 4 * lh;      // BN1 of chunk c + 1, this lane's channels
        const float* w1b = W1s + ((c + 1) & 1) * Cfg::W1_F;
        const float* w2b = W2s + ((c + 1) & 1) * Cfg::W2_F;          // (c - 1) & 1
        const float* dr = Ds + ((c + 1) & 1) * Cfg::D_F;             // D of chunk c - 1
        const float* eb = Es + (c & 1) * Cfg::E_F;
        float* en = Es + ((c + 1) & 1) * Cfg::E_F;
        float* dwr = Ds + (c & 1) * Cfg::D_F;
        const int wrow = ht * 32 + lr;
        // fragments of MFMA group g: expand K group g (A = W1 rows, B = the x fragments), then the projection's (u, Cout tile)
        auto frag = [&](int g, f32x4& av, f32x4& bv) {
            if (g < GE) {
                av = *reinterpret_cast<const f32x4*>(w1b + wrow * CIN + mid_swz<XC>(wrow, 2 * g + lh) * 4);
                bv = xf[g < KG ? g : 0];
            } else {
                const int pg = g - GE, u = pg / CT, ct = pg - u * CT;
                const int cc = 4 * wave + 2 * u + lh, row = ct * 32 + lr;
                bv = *reinterpret_cast<const f32x4*>(dr + lr * HC + mid_swz<16>(lr, cc) * 4);
                av = *reinterpret_cast<const f32x4*>(w2b + row * HC + mid_swz<16>(row, cc) * 4);
            }
        };
        // halo pixel k = 4 dy + dxx of the pair's 3 x 4 window
        auto halo = [&](int k) -> f32x4 {
            const int hs = hs00 + (k >> 2) * HWD + (k & 3);
            return *reinterpret_cast<const f32x4*>(eb + hs * HC + mid_swz<16>(hs, cq) * 4);
        };
        f32x16 acc;
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[g] = 0.f;
        f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = o0;
        f32x4 a_cur = o0, b_cur = o0, a_nxt = o0, b_nxt = o0;
        if (NG > 0) frag(0, a_cur, b_cur);
        f32x4 e_cur = halo(0), e_nxt = o0;
        // tap k = 3 dy + j is first used by slice 4 dy + j (left pixel; the right pixel uses it a slice later): it is read from
        // LDS a slice ahead of that, BN2's scale / bias in slices 10 / 11
        f32x4 tw[9], bn2s = o0, bn2b = o0;
        tw[0] = *reinterpret_cast<const f32x4*>(pt);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int sl = 0; sl < NS; ++sl) {
            // Order inside a slice (pinned): [first MFMA of the group] [the halo pixel's FMAs] | [vector-memory instructions]
            // | [LDS reads for the NEXT slice] | [the other three MFMAs] [expand epilogue piece].  Both consumers of the
            // previous slice's LDS reads come BEFORE this slice's LDS-DMA request: with a request between an LDS read and its
            // use hipcc waits lgkmcnt(0) -- for the reads it has just issued too -- and the reads stop overlapping the MFMAs.
            const int ct = sl >= GE && sl < NG ? (sl - GE) % CT : 0;
            if (UAVSAL_MID_PROBE & 1) { acc[sl & 15] += a_cur.x * b_cur.x; }
            else if (sl < GE) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.x, b_cur.x, acc, 0, 0, 0);
            else if (sl < NG) acc_o[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.x, b_cur.x, acc_o[ct], 0, 0, 0);
            // halo pixel (dy, dxx) is tap (dy, dxx) of the left pixel and (dy, dxx - 1) of the right
            if (sl < 12 && !(UAVSAL_MID_PROBE & 2)) {
                const int dy = sl >> 2, dxx = sl & 3;
                if (dxx < 3) o0 = e_cur * tw[dy * 3 + dxx] + o0;
                if (dxx > 0) o1 = e_cur * tw[dy * 3 + dxx - 1] + o1;
                // (sched_barrier pins the scheduler, not instruction selection: without this the FMAs of all twelve slices were
                // emitted in one block after the expand's last MFMA, i.e. outside the MFMAs' shadow)
                asm volatile("" : "+v"(o0), "+v"(o1));
            }
            if (sl == 12) {          // BN2, ReLU6 -> D[c & 1]
                const int op = pp_y * PW + pp_x;
                *reinterpret_cast<f32x4*>(dwr + op * HC + mid_swz<16>(op, cq) * 4) = mid_relu6(o0 * bn2s + bn2b);
                *reinterpret_cast<f32x4*>(dwr + (op + 1) * HC + mid_swz<16>(op + 1, cq) * 4) = mid_relu6(o1 * bn2s + bn2b);
            }
            __builtin_amdgcn_sched_barrier(0);
            // this slice's LDS-DMA request (every wave issues the same number per iteration)
            if (UAVSAL_MID_PROBE & 4) {}
            else if (sl < XREQ) { if (w1_more) dma_w1_one(c + 2, sl); }
            else if (sl - XREQ < W2REQ) dma_w2_one(c, sl - XREQ);
            else if (sl - XREQ == W2REQ) { if (par_more) dma_par(par_ch); }
            __builtin_amdgcn_sched_barrier(0);
            // LDS reads for the NEXT slice: they land under the three MFMAs below
            if (sl + 1 < NG && !(UAVSAL_MID_PROBE & 8)) frag(sl + 1, a_nxt, b_nxt);
            if (sl + 1 < 12 && !(UAVSAL_MID_PROBE & 2)) e_nxt = halo(sl + 1);
            if (!(UAVSAL_MID_PROBE & 2)) {
                if (sl + 1 < 12 && ((sl + 1) & 3) < 3) tw[3 * ((sl + 1) >> 2) + ((sl + 1) & 3)] = *reinterpret_cast<const f32x4*>(pt + (3 * ((sl + 1) >> 2) + ((sl + 1) & 3)) * HC);
                if (sl == 10) bn2s = *reinterpret_cast<const f32x4*>(pt + 9 * HC);
                if (sl == 11) bn2b = *reinterpret_cast<const f32x4*>(pt + 10 * HC);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (UAVSAL_MID_PROBE & 1) { acc[(sl + 1) & 15] += a_cur.y * b_cur.y + a_cur.z * b_cur.z + a_cur.w * b_cur.w; }
            else if (sl < GE) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.y, b_cur.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.z, b_cur.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.w, b_cur.w, acc, 0, 0, 0);
            } else if (sl < NG) {
                acc_o[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.y, b_cur.y, acc_o[ct], 0, 0, 0);
                acc_o[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.z, b_cur.z, acc_o[ct], 0, 0, 0);
                acc_o[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur.w, b_cur.w, acc_o[ct], 0, 0, 0);
            }
            // the expand's epilogue, two slices after its last MFMA group: relu6(bn1), zero outside the image -> E[(c + 1) & 1]
            if (EXP && (sl == GE + 1 || sl == GE + 2)) {
                const int prow = nt * 32 + lr;
#pragma unroll
                for (int q = 2 * (sl - GE - 1); q < 2 * (sl - GE - 1) + 2; ++q) {
                    f32x4 e = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
                    e = mid_relu6(e * *reinterpret_cast<const f32x4*>(pb + 8 * q) + *reinterpret_cast<const f32x4*>(pb + 64 + 8 * q));
                    if (!px_ok) e = (f32x4){0.f, 0.f, 0.f, 0.f};
                    *reinterpret_cast<f32x4*>(en + prow * HC + mid_swz<16>(prow, ht * 8 + lh + 2 * q) * 4) = e;
                }
            }
            a_cur = a_nxt; b_cur = b_nxt; e_cur = e_nxt;
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        MID_STAMP(16 + c)
        __syncthreads();
        MID_STAMP(3 + c)
    };
    if (NCH == 1) {
        iteration(std::false_type{}, std::false_type{}, 0);
    } else {
        iteration(std::true_type{}, std::false_type{}, 0);
        for (int c = 1; c + 1 < NCH; ++c) iteration(std::true_type{}, std::true_type{}, c);
        iteration(std::false_type{}, std::true_type{}, NCH - 1);
    }
    project(NCH - 1);

    // ---- the four waves' partial projections, summed in wave order; BN (linear), residual, store
    float* red = W1s;
    MID_STAMP(13)
    __syncthreads();                                             // every wave is done with the weight buffers
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int g = 0; g < 16; ++g) red[(wave * CT + ct) * 1024 + g * 64 + lane] = acc_o[ct][g];
    __syncthreads();
    // item = (patch pixel, 4 consecutive output channels 4 j ..): channel r = 4 j % 32 of tile ct sits in accumulator
    // g = 4 (r >> 3) + e of lane 32 ((r >> 2) & 1) + pixel
    for (int it = tid; it < 32 * (COUT / 4); it += 256) {
        const int px = it & 31, j = it >> 5;
        const int ct = (4 * j) >> 5, r = (4 * j) & 31;
        const int base = ct * 1024 + (4 * (r >> 3)) * 64 + ((r >> 2) & 1) * 32 + px;
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float s = red[base + e * 64];
#pragma unroll
            for (int w = 1; w < 4; ++w) s += red[w * CT * 1024 + base + e * 64];
            v[e] = s;
        }
        const int gy = oy0 + (px >> 3), gx = ox0 + (px & 7);
        if (gy >= p.H || gx >= p.W) continue;
        const size_t opix = ((size_t)n * p.H + gy) * p.W + gx;
        v = v * *reinterpret_cast<const f32x4*>(p.s2 + 4 * j) + *reinterpret_cast<const f32x4*>(p.b2 + 4 * j);
        if (p.res) v += *reinterpret_cast<const f32x4*>(p.res + opix * p.ldr + 4 * j);
        *reinterpret_cast<f32x4*>(p.out + opix * p.ldo + 4 * j) = v;
    }
    MID_STAMP(14)
}

template <int CIN, int HID, int COUT>
int launch_mid(const uavsal_fused_ir_desc* d, hipStream_t s) {
    using Cfg = MidCfg<CIN, HID, COUT>;
    MidK k;
    k.in = d->in; k.w1 = d->w1; k.s1 = d->scale1; k.b1 = d->bias1;
    k.wd = d->wd; k.sd = d->scale_d; k.bd = d->bias_d;
    k.w2 = d->w2; k.s2 = d->scale2; k.b2 = d->bias2;
    k.res = d->res; k.out = d->out;
    k.ldi = d->ldi; k.ldr = d->ldr; k.ldo = d->ldo; k.H = d->H; k.W = d->W;
    k.tiles_x = (d->W + Cfg::PW - 1) / Cfg::PW; k.tiles_y = (d->H + Cfg::PH - 1) / Cfg::PH;
    const long long nblk = (long long)d->n_img * k.tiles_y * k.tiles_x;
    if (nblk > 0x7fffffffLL) return UAVSAL_ESHAPE;
    UAVSAL_LDS_OPTIN((&fused_mid_kernel<CIN, HID, COUT>), Cfg::SMEM);
    hipLaunchKernelGGL((fused_mid_kernel<CIN, HID, COUT>), dim3((unsigned)nblk), dim3(256), Cfg::SMEM, s, k);
    return uavsal_launch_status();
}

}  // namespace

#ifdef UAVSAL_MID_STAMPS
extern "C" int uavsal_mid_stamps(unsigned long long* dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_mid_stamps), sizeof(unsigned long long) * 32);
}
#endif

// 2 when this kernel has an instance for the block (its weights then go in the conv weights' natural layouts:
// w1 [hidden][Cin], w2 [Cout][hidden]), else 0.  `launch`: run it.
int uavsal_fused_mid_dispatch(const uavsal_fused_ir_desc* d, hipStream_t s, bool launch) {
    if (!d->w1 || d->stride != 1) return launch ? UAVSAL_ESHAPE : 0;
#define UAVSAL_MID_CASE(CIN, HID, COUT)                                      \
    if (d->Cin == CIN && d->hidden == HID && d->Cout == COUT) return launch ? launch_mid<CIN, HID, COUT>(d, s) : 2;
    UAVSAL_MID_CASE(64, 384, 64)      // features.8-10; the second block of the prior nets
    UAVSAL_MID_CASE(64, 384, 96)      // features.11
    UAVSAL_MID_CASE(96, 576, 96)      // features.12, 13
    UAVSAL_MID_CASE(64, 384, 32)      // temporal sub-block of an STBlock
#undef UAVSAL_MID_CASE
    return launch ? UAVSAL_ESHAPE : 0;
}

// Fused inverted-residual block for the small-channel, HBM-bound head of the backbone:
//   out = bn3(pw_linear( relu6(bn2(dw3x3_s( relu6(bn1(pw_expand(x))) ))) )) [+ x]
// i.e. torchvision InvertedResidual / dwBlock (reference model.py:74-103) in ONE launch.  The 6x expanded
// tensors E and D never reach HBM: at 360x640 the unfused features.2 block moves 482 MB (E alone is 177 MB
// written + 177 MB read) for 40 MB of block input + output.
//
// One 256-thread workgroup produces a TY x TX patch of output pixels of one image (4x16 when the launch has few patches,
// 16x16 / 8x16 otherwise: each wave then owns PT = 4 / 2 tiles of 16 pixels and every weight fragment it loads is used
// that many times).  The hidden channels are processed in chunks of HC:
//   expand : E^T[channel][halo pixel] = W1^T . x^T on v_mfma_f32_16x16x4_f32 (exact fp32).  B = x read ONCE from global
//            memory straight into the fragment registers (lane = (halo pixel, channel group): KE contiguous channels) and
//            kept for all chunks; A = W1^T in VGPRs.  The accumulator lane holds 4 consecutive hidden channels of one
//            pixel: relu6(bn1), EXACT ZERO outside the image (the depthwise conv pads E, not x), one ds_write_b128 into
//            E[channel quad][halo slot][4] -- planes of 16 bytes per pixel, plane size a multiple of 16 slots: the 16 lanes
//            one LDS cycle of a ds_read_b128 serves ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md, LDS) are 8 + 8 pixels of
//            two neighbouring planes and cover the 64 banks exactly when the pixels are consecutive slots, which they are
//            (patches are 16 outputs wide; for stride 2 the halo columns are stored even columns first, then odd);
//   dw     : lane = (output pixel l & 15, channel group l >> 4 of HC/4 channels): nine taps x HC/16 ds_read_b128, tap
//            weights / BN in VGPRs, relu6(bn2) -- and the lane's values ARE its B fragment of the projection MFMA
//            (k index (l >> 4, s) <-> channel (l >> 4) * HC/4 + s), so D never goes through LDS;
//   project: OUT^T[cout][pixel] += W2^T . D^T on the same MFMA, accumulators live across the chunks; the lane ends with 4
//            consecutive output channels of one pixel: bn3, optional residual, one 16-byte store.
// Two buffers of E when they fit (one barrier per chunk), else one (two barriers).
// The first version of this kernel (round 2) kept E channel-major with scalar LDS accesses (4-way bank conflicts on
// both sides), sent D through LDS, reloaded every weight fragment per 16 output pixels and staged x in LDS:
// profiles/r3_fused_ir.md has the before / after.
// MFMA lane maps (cdna_hip_programming.md): A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15],
// D[i = 4 * (l >> 4) + reg][j = l & 15].
// This kernel is used for every GEMM precision mode: it is exact fp32 and the blocks it covers are
// bandwidth-bound (4 GMAC per clip in total).
#include "common.h"

namespace {

struct FusedK {
    const float* in; const float* w1; const float* s1; const float* b1;
    const float* wd; const float* sd; const float* bd;
    const float* w2; const float* s2; const float* b2;
    const float* res; float* out;
    int ldi, ldr, ldo, H, W, Ho, Wo, tiles_x, tiles_y;
};

typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x4 relu6_4(f32x4 v) {
    return (f32x4){__builtin_amdgcn_fmed3f(v.x, 0.f, 6.f), __builtin_amdgcn_fmed3f(v.y, 0.f, 6.f), __builtin_amdgcn_fmed3f(v.z, 0.f, 6.f),
                   __builtin_amdgcn_fmed3f(v.w, 0.f, 6.f)};
}

#ifndef UAVSAL_FUSED_STAGE_W
#define UAVSAL_FUSED_STAGE_W 1      /* 0: every wave loads every chunk's weights from global memory (the round-3 kernel) */
#endif
#ifndef UAVSAL_FUSED_NBUF2_MAX
#define UAVSAL_FUSED_NBUF2_MAX (80 * 1024)     /* two E buffers (one barrier per chunk) when they fit, with the staged weights, in this many bytes */
#endif

template <int S, int TX, int PT>
struct FusedGeom {
    static constexpr int RP = 16 / TX;                          // output rows per 16-pixel tile
    static constexpr int TY = 4 * PT * RP;
    static constexpr int IH = (TY - 1) * S + 3, IW = (TX - 1) * S + 3;
    static constexpr int HALF = (IW + 1) / 2;
    static constexpr int IWP = S == 2 ? 2 * HALF : IW;          // slots per halo row (stride 2: even columns, then odd)
    static constexpr int NSLOT = IH * IWP, NRT = (NSLOT + 15) / 16;
};

template <int CIN, int HID, int COUT, int S, bool EXPAND, int HC, int TX, int PT>
struct FusedCfg : FusedGeom<S, TX, PT> {
    using G = FusedGeom<S, TX, PT>;
    // E in LDS, float offset of (channel quad qp, halo slot): with an expand conv, planes [qp][slot][4]; without one (the
    // staged input: 8 lanes write the 128 contiguous bytes of a pixel) pixel-major rows of HC + 4 floats
    static constexpr int PLANE = EXPAND ? G::NRT * 16 * 4 : 4;
    static constexpr int SLOTF = EXPAND ? 4 : HC + 4;
    static constexpr int EBUF = EXPAND ? (HC / 4) * PLANE : G::NRT * 16 * SLOTF;   // floats per E buffer
    static constexpr int NCH = HID / HC;
    // Round 4: a chunk's weights are staged in LDS (two buffers, filled a chunk ahead: global loads at the top of chunk ch, LDS
    // writes in front of its barrier) and every fragment / per-lane constant is read from there.  Loaded from global memory per
    // chunk and wave they were 27-40 vector-memory instructions per chunk and wave -- 166-465 per wave and launch
    // (`SQ_INSTS_VMEM`, profiles/r4_sq_pmc_f32_c1.md) -- and the CU's vector-memory issue (one address-coalescing pipe for its
    // four SIMDs), not the matrix pipe, set the pace of a chunk.  (First form, kept until the stamps of tools/fir_stamps.py: the
    // WHOLE block's weights copied once per workgroup -- 20-82 KB, so only the stride-1 instances could afford it.)
    // Chunk layout (floats): w1 [CIN][HC] | s1 [HC] | b1 [HC] | wd [9][HC] | sd [HC] | bd [HC] | w2 [HC][COUT].
    static constexpr int CW_S1 = CIN * HC, CW_B1 = CW_S1 + HC, CW_WD = CW_B1 + HC, CW_SD = CW_WD + 9 * HC, CW_BD = CW_SD + HC,
                         CW_W2 = CW_BD + HC, CW_F = CW_W2 + HC * COUT;
    static constexpr bool STAGE_W = EXPAND && UAVSAL_FUSED_STAGE_W;
    static constexpr int NBUF = (EXPAND && NCH > 1 && (2 * EBUF + (STAGE_W ? 2 * CW_F : 0)) * 4 <= UAVSAL_FUSED_NBUF2_MAX) ? 2 : 1;
    static constexpr size_t E_BYTES = (size_t)NBUF * EBUF * 4;
    static constexpr int WBASE = NBUF * EBUF;                   // float offset of the staged weights
    static constexpr size_t SMEM = E_BYTES + (STAGE_W ? (size_t)2 * CW_F * 4 : 0);
};

#ifndef UAVSAL_FUSED_PROBE
#define UAVSAL_FUSED_PROBE 0      /* timing experiments only (tools/fused_probe.py): 1 no expand MFMA, 2 no depthwise taps,
                                     4 no projection MFMA, 8 every chunk uses chunk 0's weights, 16 no barriers, 32 no E writes */
#endif

#ifdef UAVSAL_FIR_STAMPS      // diagnostic build only (tools/fir_stamps.py): s_memtime at the phase boundaries of a mid-grid workgroup, wave 0,
                              // every stamp behind a full s_waitcnt (the phases are serialised by the stamps: costs, not overlap)
__device__ unsigned long long g_fir_stamps[128];
#define FIR_STAMP(i) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); \
                       if (blockIdx.x == gridDim.x / 2 && tid == 0) g_fir_stamps[i] = __builtin_amdgcn_s_memtime(); }
#else
#define FIR_STAMP(i)
#endif

template <int CIN, int HID, int COUT, int S, bool EXPAND, int HC, int TX, int PT>
__global__ __launch_bounds__(256, 2) void fused_ir_kernel(const FusedK p) {
    using Cfg = FusedCfg<CIN, HID, COUT, S, EXPAND, HC, TX, PT>;
    constexpr int RP = Cfg::RP, TY = Cfg::TY, IH = Cfg::IH, IW = Cfg::IW, HALF = Cfg::HALF, IWP = Cfg::IWP;
    constexpr int NSLOT = Cfg::NSLOT, NRT = Cfg::NRT, RTW = (NRT + 3) / 4;
    constexpr int CPL = HC / 4, CQ = CPL / 4, PLANE = Cfg::PLANE, SLOTF = Cfg::SLOTF;   // channels per lane in the depthwise phase (= k steps)
    constexpr int NCH = Cfg::NCH, NCTE = HC / 16, NCT = (COUT + 15) / 16, KE = CIN / 4;
    constexpr int EBUF = Cfg::EBUF, NBUF = Cfg::NBUF;
    static_assert(HID % HC == 0 && HC % 16 == 0 && CIN % 8 == 0 && COUT % 4 == 0, "channel blocking");
    static_assert(EXPAND || (HID == CIN && NCH == 1), "without an expand conv the hidden tensor is the input");
    static_assert(KE % 2 == 0, "x fragments are loaded 8 bytes at a time");

    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    int t = blockIdx.x;
    const int tx = t % p.tiles_x; t /= p.tiles_x;
    const int ty = t % p.tiles_y;
    const int n = t / p.tiles_y;
    const int oy0 = ty * TY, ox0 = tx * TX;
    const int iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;
    const float* inb = p.in + (size_t)n * p.H * p.W * p.ldi;
    FIR_STAMP(0)

    // halo slot -> input pixel; false for the pad slots
    auto slot_pixel = [&](int slot, int& gy, int& gx) -> bool {
        const int iy = slot / IWP, rem = slot - iy * IWP;
        int ix = rem;
        if (S == 2) { const int par = rem >= HALF ? 1 : 0; ix = 2 * (rem - par * HALF) + par; }
        gy = iy0 + iy; gx = ix0 + ix;
        return slot < NSLOT && ix < IW && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
    };

    auto slot_read = [&](int slot) -> bool {          // does the depthwise phase ever read this slot
        const int iy = slot / IWP, rem = slot - iy * IWP;
        return slot < NSLOT && (S == 2 ? rem != IWP - 1 || (IW & 1) == 0 : true);
    };

    // ---- the x fragments of this wave's halo row tiles (round-robin over the 4 waves), kept for every chunk ----
    float xv[EXPAND ? RTW : 1][EXPAND ? KE : 1];
    unsigned outside = 0;                      // bit j: this lane's pixel of row tile j lies outside the image (E must be 0 there)
    unsigned any_outside = 0;                  // bit j, wave-uniform: some lane of row tile j does (interior patches: none)
    if (EXPAND) {
#pragma unroll
        for (int j = 0; j < RTW; ++j) {
            const int rt = wave + 4 * j;
            int gy, gx;
            const bool in = rt < NRT && slot_pixel(rt * 16 + l15, gy, gx);
            const bool zero = !in && rt < NRT && slot_read(rt * 16 + l15);   // the pad slots are never read: no zero needed
            outside |= (zero ? 1u : 0u) << j;
            any_outside |= (__builtin_amdgcn_ballot_w64(zero) != 0ull ? 1u : 0u) << j;
            const float* src = inb + ((size_t)(in ? gy : 0) * p.W + (in ? gx : 0)) * p.ldi + lq * KE;
            if (KE % 4 == 0) {
#pragma unroll
                for (int s = 0; s < KE; s += 4) {
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (in) v = *reinterpret_cast<const f32x4*>(src + s);
                    xv[j][s] = v.x; xv[j][s + 1] = v.y; xv[j][s + 2] = v.z; xv[j][s + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int s = 0; s < KE; s += 2) {
                    f32x2 v = {0.f, 0.f};
                    if (in) v = *reinterpret_cast<const f32x2*>(src + s);
                    xv[j][s] = v.x; xv[j][s + 1] = v.y;
                }
            }
        }
    } else {
        // the input IS the hidden tensor: stage the halo patch pixel-major (zero outside the image and in the pad slots);
        // all the loads of a thread first, then its LDS writes
        constexpr int NQ = CIN / 4, NST = (NRT * 16 * NQ + 255) / 256;
        f32x4 st[NST];
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int idx = tid + 256 * i;
            const int slot = idx / NQ, q = idx - slot * NQ;
            int gy, gx;
            st[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (idx < NRT * 16 * NQ && slot_pixel(slot, gy, gx)) st[i] = *reinterpret_cast<const f32x4*>(inb + ((size_t)gy * p.W + gx) * p.ldi + q * 4);
        }
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int idx = tid + 256 * i;
            const int slot = idx / NQ, q = idx - slot * NQ;
            if (idx < NRT * 16 * NQ) *reinterpret_cast<f32x4*>(lds + (size_t)q * PLANE + slot * SLOTF) = st[i];
        }
    }

    FIR_STAMP(1)
    f32x4 acc_o[PT][NCT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) acc_o[pt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- weight staging: items of 16 bytes; item -> (source address in chunk 0, floats between chunks, offset in the chunk block)
    constexpr bool STAGE_W = Cfg::STAGE_W;
    constexpr int HQ = HC / 4, NI_W1 = CIN * HQ, NI_V = HQ, NI_WD = 9 * HQ, NI_W2 = HC * COUT / 4;
    constexpr int NITEM = NI_W1 + 2 * NI_V + NI_WD + 2 * NI_V + NI_W2, WROUNDS = (NITEM + 255) / 256;
    float* const wl = lds + Cfg::WBASE;
    const float* wsrc[STAGE_W ? WROUNDS : 1];
    int wdst[STAGE_W ? WROUNDS : 1], wstep[STAGE_W ? WROUNDS : 1];
    f32x4 wreg[STAGE_W ? WROUNDS : 1];
    if (STAGE_W) {
#pragma unroll
        for (int r = 0; r < WROUNDS; ++r) {
            int it = tid + 256 * r;
            wsrc[r] = nullptr; wdst[r] = 0; wstep[r] = HC;
            if (it < NI_W1) { const int k = it / HQ, q = it - k * HQ; wsrc[r] = p.w1 + (size_t)k * HID + 4 * q; wdst[r] = k * HC + 4 * q; }
            else if ((it -= NI_W1) < NI_V) { wsrc[r] = p.s1 + 4 * it; wdst[r] = Cfg::CW_S1 + 4 * it; }
            else if ((it -= NI_V) < NI_V) { wsrc[r] = p.b1 + 4 * it; wdst[r] = Cfg::CW_B1 + 4 * it; }
            else if ((it -= NI_V) < NI_WD) { const int tp = it / HQ, q = it - tp * HQ; wsrc[r] = p.wd + (size_t)tp * HID + 4 * q; wdst[r] = Cfg::CW_WD + tp * HC + 4 * q; }
            else if ((it -= NI_WD) < NI_V) { wsrc[r] = p.sd + 4 * it; wdst[r] = Cfg::CW_SD + 4 * it; }
            else if ((it -= NI_V) < NI_V) { wsrc[r] = p.bd + 4 * it; wdst[r] = Cfg::CW_BD + 4 * it; }
            else if ((it -= NI_V) < NI_W2) { wsrc[r] = p.w2 + 4 * it; wdst[r] = Cfg::CW_W2 + 4 * it; wstep[r] = HC * COUT; }
        }
#pragma unroll
        for (int r = 0; r < WROUNDS; ++r)
            if (wsrc[r]) *reinterpret_cast<f32x4*>(wl + wdst[r]) = *reinterpret_cast<const f32x4*>(wsrc[r]);      // chunk 0
        __syncthreads();
    }
    const float* const g_w1 = p.w1;
    FIR_STAMP(2)

    for (int ch = 0; ch < NCH; ++ch) {
        const int c0 = (UAVSAL_FUSED_PROBE & 8) ? 0 : ch * HC;
        const float* const wc = wl + (ch & 1) * Cfg::CW_F;             // this chunk's staged weights
        // ---- this chunk's weights, as MFMA A fragments / per-lane constants (global loads, cache-resident) ----
        float w1f[EXPAND ? NCTE : 1][EXPAND ? KE : 1];
        f32x4 s1q[EXPAND ? NCTE : 1], b1q[EXPAND ? NCTE : 1];
        if (EXPAND) {
#pragma unroll
            for (int ct = 0; ct < NCTE; ++ct) {
#pragma unroll
                for (int s = 0; s < KE; ++s)
                    w1f[ct][s] = STAGE_W ? wc[(lq * KE + s) * HC + ct * 16 + l15] : g_w1[(size_t)(lq * KE + s) * HID + c0 + ct * 16 + l15];
                s1q[ct] = STAGE_W ? *reinterpret_cast<const f32x4*>(wc + Cfg::CW_S1 + ct * 16 + 4 * lq)
                                  : *reinterpret_cast<const f32x4*>(p.s1 + c0 + ct * 16 + 4 * lq);
                b1q[ct] = STAGE_W ? *reinterpret_cast<const f32x4*>(wc + Cfg::CW_B1 + ct * 16 + 4 * lq)
                                  : *reinterpret_cast<const f32x4*>(p.b1 + c0 + ct * 16 + 4 * lq);
            }
        }
        f32x4 wdq[9][CQ], sdq[CQ], bdq[CQ];
#pragma unroll
        for (int q = 0; q < CQ; ++q) {
#pragma unroll
            for (int tp = 0; tp < 9; ++tp)
                wdq[tp][q] = STAGE_W ? *reinterpret_cast<const f32x4*>(wc + Cfg::CW_WD + tp * HC + lq * CPL + 4 * q)
                                     : *reinterpret_cast<const f32x4*>(p.wd + (size_t)tp * HID + c0 + lq * CPL + 4 * q);
            sdq[q] = STAGE_W ? *reinterpret_cast<const f32x4*>(wc + Cfg::CW_SD + lq * CPL + 4 * q) : *reinterpret_cast<const f32x4*>(p.sd + c0 + lq * CPL + 4 * q);
            bdq[q] = STAGE_W ? *reinterpret_cast<const f32x4*>(wc + Cfg::CW_BD + lq * CPL + 4 * q) : *reinterpret_cast<const f32x4*>(p.bd + c0 + lq * CPL + 4 * q);
        }
        float w2f[NCT][CPL];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const int co = ct * 16 + l15;
#pragma unroll
            for (int s = 0; s < CPL; ++s)
                w2f[ct][s] = co < COUT ? (STAGE_W ? wc[Cfg::CW_W2 + (lq * CPL + s) * COUT + co] : p.w2[(size_t)(c0 + lq * CPL + s) * COUT + co]) : 0.f;
        }
        if (STAGE_W && ch + 1 < NCH) {           // the next chunk's weights: global loads now, LDS writes in front of this chunk's barrier
#pragma unroll
            for (int r = 0; r < WROUNDS; ++r)
                if (wsrc[r]) wreg[r] = *reinterpret_cast<const f32x4*>(wsrc[r] + (size_t)(ch + 1) * wstep[r]);
        }

        FIR_STAMP(8 + 5 * ch)
        float* eb = lds + (NBUF == 2 ? (ch & 1) * EBUF : 0);
        if (EXPAND) {
            if (NBUF == 1 && ch > 0 && !(UAVSAL_FUSED_PROBE & 16)) __syncthreads();       // the previous chunk's depthwise reads are done
#pragma unroll
            for (int j = 0; j < RTW; ++j) {
                const int rt = wave + 4 * j;
                if (4 * j + 3 >= NRT && rt >= NRT) continue;   // only the last round can run past the patch (wave-uniform)
                f32x4 e[NCTE];
#pragma unroll
                for (int ct = 0; ct < NCTE; ++ct) {
                    e[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < KE; ++s) {
                        if (UAVSAL_FUSED_PROBE & 1) e[ct][s & 3] += w1f[ct][s] * xv[j][s];
                        else e[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(w1f[ct][s], xv[j][s], e[ct], 0, 0, 0);
                    }
                    e[ct] = relu6_4(e[ct] * s1q[ct] + b1q[ct]);
                }
                if ((any_outside >> j) & 1u) {               // zero padding of the depthwise conv: E = 0 outside the image
                    const bool z = (outside >> j) & 1u;
#pragma unroll
                    for (int ct = 0; ct < NCTE; ++ct) if (z) e[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int ct = 0; ct < NCTE; ++ct)
                    if (!(UAVSAL_FUSED_PROBE & 32) || e[ct].x == 123.f) *reinterpret_cast<f32x4*>(eb + (size_t)(ct * 4 + lq) * PLANE + (rt * 16 + l15) * SLOTF) = e[ct];
            }
        }
        if (STAGE_W && ch + 1 < NCH) {
#pragma unroll
            for (int r = 0; r < WROUNDS; ++r)
                if (wsrc[r]) *reinterpret_cast<f32x4*>(wl + ((ch + 1) & 1) * Cfg::CW_F + wdst[r]) = wreg[r];
        }
        FIR_STAMP(9 + 5 * ch)
        if (!(UAVSAL_FUSED_PROBE & 16)) __syncthreads();          // E (or the staged input) and the next chunk's weights visible
        FIR_STAMP(10 + 5 * ch)
        // ---- depthwise 3x3 (stride S) + BN + ReLU6, then the projection with the lane's values as B fragment ----
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            const int g = wave * PT + pt;
            const int orow = g * RP + l15 / TX, ocol = l15 % TX;
            f32x4 dq[CQ];
#pragma unroll
            for (int q = 0; q < CQ; ++q) dq[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ky = 0; ky < ((UAVSAL_FUSED_PROBE & 2) ? 1 : 3); ++ky)
#pragma unroll
                for (int kx = 0; kx < ((UAVSAL_FUSED_PROBE & 2) ? 1 : 3); ++kx) {
                    const int slot = (orow * S + ky) * IWP + (S == 2 ? (kx & 1) * HALF + ocol + (kx >> 1) : ocol + kx);
                    const float* e = eb + (size_t)(lq * CQ) * PLANE + slot * SLOTF;
#pragma unroll
                    for (int q = 0; q < CQ; ++q) dq[q] = *reinterpret_cast<const f32x4*>(e + (size_t)q * PLANE) * wdq[ky * 3 + kx][q] + dq[q];
                }
#pragma unroll
            for (int q = 0; q < CQ; ++q) dq[q] = relu6_4(dq[q] * sdq[q] + bdq[q]);
            if (pt == PT - 1) { FIR_STAMP(11 + 5 * ch) }
#pragma unroll
            for (int s = 0; s < CPL; ++s)
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
                    if (UAVSAL_FUSED_PROBE & 4) acc_o[pt][ct][s & 3] += w2f[ct][s] * dq[s >> 2][s & 3];
                    else acc_o[pt][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(w2f[ct][s], dq[s >> 2][s & 3], acc_o[pt][ct], 0, 0, 0);
        }
        FIR_STAMP(12 + 5 * ch)
    }
    FIR_STAMP(3)

    // ---- epilogue: BN (linear), residual, store: lane holds output channels ct*16 + 4*lq .. +3 of pixel l15 of its tiles ----
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        const int co = ct * 16 + 4 * lq;
        if (co >= COUT) continue;
        const f32x4 sc = *reinterpret_cast<const f32x4*>(p.s2 + co), bi = *reinterpret_cast<const f32x4*>(p.b2 + co);
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            const int g = wave * PT + pt;
            const int gy = oy0 + g * RP + l15 / TX, gx = ox0 + l15 % TX;
            if (gy >= p.Ho || gx >= p.Wo) continue;
            const size_t opix = ((size_t)n * p.Ho + gy) * p.Wo + gx;
            f32x4 v = acc_o[pt][ct] * sc + bi;
            if (p.res) v += *reinterpret_cast<const f32x4*>(p.res + opix * p.ldr + co);
            *reinterpret_cast<f32x4*>(p.out + opix * p.ldo + co) = v;
        }
    }
    FIR_STAMP(4)
}

template <int CIN, int HID, int COUT, int S, bool EXPAND, int HC, int TX, int PT>
int launch_fused(const uavsal_fused_ir_desc* d, hipStream_t s) {
    using Cfg = FusedCfg<CIN, HID, COUT, S, EXPAND, HC, TX, PT>;
    FusedK k;
    k.in = d->in; k.w1 = d->w1; k.s1 = d->scale1; k.b1 = d->bias1;
    k.wd = d->wd; k.sd = d->scale_d; k.bd = d->bias_d;
    k.w2 = d->w2; k.s2 = d->scale2; k.b2 = d->bias2;
    k.res = d->res; k.out = d->out;
    k.ldi = d->ldi; k.ldr = d->ldr; k.ldo = d->ldo; k.H = d->H; k.W = d->W;
    k.Ho = (d->H - 1) / S + 1; k.Wo = (d->W - 1) / S + 1;
    k.tiles_x = (k.Wo + TX - 1) / TX; k.tiles_y = (k.Ho + Cfg::TY - 1) / Cfg::TY;
    const long long nblk = (long long)d->n_img * k.tiles_y * k.tiles_x;
    if (nblk > 0x7fffffffLL) return UAVSAL_ESHAPE;
    UAVSAL_LDS_OPTIN((&fused_ir_kernel<CIN, HID, COUT, S, EXPAND, HC, TX, PT>), Cfg::SMEM);
    hipLaunchKernelGGL((fused_ir_kernel<CIN, HID, COUT, S, EXPAND, HC, TX, PT>), dim3((unsigned)nblk), dim3(256), Cfg::SMEM, s, k);
    return uavsal_launch_status();
}

// Patch shape: the big one (16x16 outputs at stride 1, 8x16 at stride 2: fewer halo pixels, every weight fragment used
// 4 / 2 times) when it still gives UAVSAL_FUSED_BIG_MIN workgroups, else 4x16.  A function of the shape only.
#ifndef UAVSAL_FUSED_BIG_MIN
#define UAVSAL_FUSED_BIG_MIN 400
#endif
template <int CIN, int HID, int COUT, int S, bool EXPAND, int HCS, int HCB>
int launch_fused_shape(const uavsal_fused_ir_desc* d, hipStream_t s) {
    constexpr bool BIG_AUTO = !(CIN == 32 && COUT == 64);     // features.7 (80 x-fragment registers per wave in the big patch): the small patch measured faster at every size
    constexpr int BPT = S == 1 ? 4 : 2;                       // 16 x 16 or 8 x 16 outputs
    constexpr int BTY = 4 * BPT;
    const int Ho = (d->H - 1) / S + 1, Wo = (d->W - 1) / S + 1;
    const long long big = (long long)d->n_img * ((Ho + BTY - 1) / BTY) * ((Wo + 15) / 16);
    static const long long big_min = [] { const char* e = getenv("UAVSAL_FUSED_BIG_MIN"); return e ? atoll(e) : (long long)UAVSAL_FUSED_BIG_MIN; }();
    const bool use_big = d->tile == 2 || (d->tile == 0 && BIG_AUTO && big >= big_min);
    return use_big ? launch_fused<CIN, HID, COUT, S, EXPAND, HCB, 16, BPT>(d, s) : launch_fused<CIN, HID, COUT, S, EXPAND, HCS, 16, 1>(d, s);
}

// the channel / stride combinations that exist as instances: MobileNetV2 features[1..7] (model_feature.py:62-66)
int dispatch(const uavsal_fused_ir_desc* d, hipStream_t s, bool launch) {
#define UAVSAL_FUSED_UNPAREN(...) __VA_ARGS__
#define UAVSAL_FUSED_CASE(CIN, HID, COUT, S, EXP, HCC)                                                   \
    if (d->Cin == CIN && d->hidden == HID && d->Cout == COUT && d->stride == S && (d->w1 != nullptr) == EXP) \
        return launch ? launch_fused_shape<CIN, HID, COUT, S, EXP, UAVSAL_FUSED_UNPAREN HCC>(d, s) : 1;
#ifndef UAVSAL_FUSED_HCS
#define UAVSAL_FUSED_HCS 16       /* hidden chunk of the blocks with an expand conv, 4x16 patches */
#endif
#ifndef UAVSAL_FUSED_HCS144
#define UAVSAL_FUSED_HCS144 16    /* ... hidden = 144 (16 or 48) */
#endif
#ifndef UAVSAL_FUSED_HCB
#define UAVSAL_FUSED_HCB 16       /* ... big patches */
#endif
#ifndef UAVSAL_FUSED_HCB144
#define UAVSAL_FUSED_HCB144 16
#endif
    UAVSAL_FUSED_CASE(32, 32, 16, 1, false, (32, 32))     // features.1  (t = 1: no expand conv)
    UAVSAL_FUSED_CASE(16, 96, 24, 2, true, (UAVSAL_FUSED_HCS, UAVSAL_FUSED_HCB))      // features.2
    UAVSAL_FUSED_CASE(24, 144, 24, 1, true, (UAVSAL_FUSED_HCS144, UAVSAL_FUSED_HCB144))     // features.3
    UAVSAL_FUSED_CASE(24, 144, 32, 2, true, (UAVSAL_FUSED_HCS144, UAVSAL_FUSED_HCB144))     // features.4
    UAVSAL_FUSED_CASE(32, 192, 32, 1, true, (UAVSAL_FUSED_HCS, UAVSAL_FUSED_HCB))     // features.5, features.6
    UAVSAL_FUSED_CASE(32, 192, 64, 2, true, (UAVSAL_FUSED_HCS, UAVSAL_FUSED_HCB))     // features.7
#undef UAVSAL_FUSED_CASE
    return launch ? UAVSAL_ESHAPE : 0;
}

}  // namespace

// fused_mid.hip: the GEMM-shaped kernel for the mid-channel blocks (Cin 64 / 96); 2 = instance exists
int uavsal_fused_mid_dispatch(const uavsal_fused_ir_desc* d, hipStream_t s, bool launch);

#ifdef UAVSAL_FIR_STAMPS
extern "C" int uavsal_fir_stamps(unsigned long long* dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_fir_stamps), sizeof(unsigned long long) * 128);
}
#endif

extern "C" int uavsal_fused_ir_supported(const uavsal_fused_ir_desc* d) {
    if (!d) return 0;
    const int mid = uavsal_fused_mid_dispatch(d, nullptr, false);
    return mid ? mid : dispatch(d, nullptr, false);
}

extern "C" int uavsal_fused_ir(const uavsal_fused_ir_desc* d, uavsal_stream_t stream) {
    if (!d || !d->in || !d->wd || !d->scale_d || !d->bias_d || !d->w2 || !d->scale2 || !d->bias2 || !d->out) return UAVSAL_EINVAL;
    if (d->w1 && (!d->scale1 || !d->bias1)) return UAVSAL_EINVAL;
    if (d->n_img <= 0 || d->H <= 0 || d->W <= 0) return UAVSAL_EINVAL;
    if ((d->ldi & 3) || d->ldi < d->Cin || (d->ldo & 3) || d->ldo < d->Cout || !uavsal_aligned16(d->in) || !uavsal_aligned16(d->out) ||
        !uavsal_aligned16(d->wd) || !uavsal_aligned16(d->scale_d) || !uavsal_aligned16(d->bias_d) || !uavsal_aligned16(d->scale2) ||
        !uavsal_aligned16(d->bias2) || (d->w1 && (!uavsal_aligned16(d->scale1) || !uavsal_aligned16(d->bias1))) ||
        (d->res && ((d->ldr & 3) || !uavsal_aligned16(d->res))))
        return UAVSAL_EALIGN;
    if ((d->w1 && !uavsal_aligned16(d->w1)) || !uavsal_aligned16(d->w2)) return UAVSAL_EALIGN;      // staged in 16-byte pieces
    if (d->tile < 0 || d->tile > 2) return UAVSAL_EINVAL;
    if (d->res && (d->ldr < d->Cout || d->stride != 1 || d->Cin != d->Cout)) return UAVSAL_ESHAPE;
    if (uavsal_fused_mid_dispatch(d, nullptr, false)) return uavsal_fused_mid_dispatch(d, (hipStream_t)stream, true);
    return dispatch(d, (hipStream_t)stream, true);
}

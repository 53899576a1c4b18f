// Depthwise 3x3 -> projection in one launch: its own translation unit (split out of conv_gemm.hip in round 4 so that
// the kernel can be rebuilt and probed on its own).
#include "conv_gemm_common.h"

#ifndef UAVSAL_DWPROJ_TEPI
#define UAVSAL_DWPROJ_TEPI 0      /* 1: transposed accumulators + 16-byte stores straight from registers instead of the staged epilogue
                                     (32-row blocks through LDS, eight barriers per tile).  Built in round 4, parity-green, NOT the default:
                                     a wave store then covers 32 pixels x 32 bytes instead of whole rows -- fp32 one clip 4.390-4.400 vs
                                     4.385-4.395 ms (st.sp 230 -> 228 us, fucb 68 -> 72), f16x3 eight clips 18.56 vs 18.08 ms */
#endif

namespace {

// =====================================================================================
// Depthwise 3x3 + BN + ReLU6 -> 1x1 projection + BN (+ residual) in ONE launch, fp32, LDS halo tile
// (dwBlock, reference model.py:92-95): D = relu6(bn(dw3x3(E))) never reaches HBM.
//
// A workgroup owns an 8 x 16 pixel patch of one image (= the GEMM's 128-row M tile) and BN output channels,
// and walks K in steps of 16 hidden channels.  Per step it
//   * requests by LDS-DMA, four steps ahead: the 10 x 18 halo of E for those channels (64 B per pixel, rows
//     padded to 19 slots; out-of-image pixels read a zero page = the convolution's zero padding), the step's
//     depthwise taps / BN scale / BN bias (11 rows x 64 B, straight from the [9][C] tap-major array) and the
//     BN x 64 B projection-weight panel.  The requests are issued by the SECOND wave of every SIMD (waves 4-7),
//     the same number by each (spare ones fetch the zero page into a scratch KB: the counted vmcnt is then one
//     immediate), one behind each of its MFMA groups, with running per-lane source pointers: no branches, no
//     address arithmetic beyond one 64-bit add per request;
//   * computes the depthwise two steps ahead of the multiply, on the FIRST wave of every SIMD (waves 0-3, the
//     older ones, at raised priority): a lane owns a 1 x 2 pixel strip x 4 channels, reads its 3 x 4 halo pixels
//     (12 ds_read_b128 at immediate offsets from one base, conflict-free with the 19-slot row pitch) plus
//     the 11 weight rows and writes two A-tile rows, while the SIMD's other wave keeps the matrix pipe busy;
//   * multiplies the step whose fragments were read from LDS BEFORE the barrier (fragments are loaded one step
//     ahead, half of them after the first half of the MFMAs, the other half after the last MFMA: those reads
//     stay in flight ACROSS the barrier -- it only waits for the older LDS accesses -- which is why there is one
//     more weight panel and A tile than the requests need).  Between a wave's last MFMA of a step and its first
//     of the next there is nothing but that wait and the barrier: with one workgroup per CU every cycle spent
//     there idles the matrix pipe (4096 MFMA cycles per step and SIMD).
// One barrier per K step; E slots x3, weight panels x5, A tiles x3.  A-tile row r holds the pixel
// (y, x) = (4 r5 + r[2:1], 8 r6 + 2 (r0 + 2 r4) + r3) (r_i = bit i of r), which makes the depthwise stores of a
// lane group fall on distinct banks; the epilogue inverts it.
// PREC = F16X3: the same walk with the split-fp16 product.  The depthwise phase writes the A tile as
// [hi 16 halves | lo 16 halves] per row (hi = fp16_rtz(16 d), lo = fp16_rtz(16 d - hi): the split-shadow values), the
// weight panel arrives pre-split the same way ('f16x3j': [Cin/16][Npad][hi 16 | lo 16]), and a K step is
// lo*hi + hi*lo + hi*hi on v_mfma_f32_32x32x16_f16 -- 12 instead of 32 MFMA issues per wave, so these launches are
// bound by the requests (29 KB per step and workgroup), not by the matrix pipe.  All fragments of the next step are
// read after the last MFMA and stay in flight across the barrier.
// Shapes: stride 1, dilation 1, hidden channels % 16 == 0 (host: dwproj_eligible).
// PWV = 4 (round 4, "producer waves"): four EXTRA waves, one per SIMD, do nothing but the depthwise and the requests; the
// WAVES_M x WAVES_N MFMA waves do nothing but fragment reads and MFMAs (three waves per SIMD, <= 168 VGPRs each).  The
// depthwise VALU / LDS work and the request issue then never sit in an MFMA wave's in-order instruction stream.
template <int PREC, int WAVES_M, int WAVES_N, int WM, int WN, int PWV = 0>
__global__ __launch_bounds__((WAVES_M * WAVES_N + PWV) * 64, PWV ? 3 : 1) void dwproj_kernel(const ConvK p) {
    static_assert(PREC == UAVSAL_PREC_F32 || PREC == UAVSAL_PREC_F16X3, "fp32 or split-fp16");
    constexpr bool H16 = PREC == UAVSAL_PREC_F16X3;
    constexpr int PH = 8, PW = 16, HPITCH = PW + 3, NHSLOT = (PH + 2) * HPITCH;
    constexpr int BM = PH * PW, BN = WAVES_N * WN * 32;
    constexpr int NW = WAVES_M * WAVES_N, NT = (NW + PWV) * 64;      // NW: MFMA waves; NT: all threads of the workgroup
    constexpr int KT = 16, DIST = 4;
    constexpr bool TEPI = UAVSAL_DWPROJ_TEPI != 0;
    static_assert(PWV == 0 || (PWV == 4 && NW == 8), "producer waves: 8 MFMA waves + 4");
    constexpr int E_REQ = (NHSLOT + 15) / 16;         // 12 wave requests of 16 halo slots, then one for the dw weights
    constexpr int W_OFF = E_REQ * 1024;               // [9 taps | scale | bias][16 channels] behind the halo
    constexpr int E_SLOT = W_OFF + 1024, NE = 3;
    constexpr int B_REQ = BN / 16, B_SLOT = BN * 64, NB = 5;
    constexpr int A_SLOT = BM * 64, NA = 3;
    constexpr int NLW = 4;                            // waves that issue the requests: the second wave of each SIMD (8-wave
                                                      // instances; the first one runs the depthwise), else all four
    constexpr int E_IT = (E_REQ + 1 + NLW - 1) / NLW, B_IT = (B_REQ + NLW - 1) / NLW, NREQ = E_IT + B_IT;
    constexpr int NGRP = (H16 ? 1 : 2) * WM * WN, RPG = (NREQ + NGRP - 1) / NGRP;
    // fragment reads that may stay in flight across the barrier: with three A tiles everything issued after a step's
    // last MFMA, with two only the (youngest) weight-panel reads -- the next depthwise overwrites the A tile just read
    constexpr int NTAIL = NA == 3 ? (H16 ? 2 : 1) * (WM + WN) : WN;
    constexpr int DW_ITEMS = BM * 4 / 2 / 64;         // waves' worth of (1 x 2 strip, 4 channels) items: 4
    static_assert(WAVES_M * WM * 32 == BM, "the M tile is the 8 x 16 patch");
    static_assert(2 * B_SLOT >= 32 * BN * 4, "epilogue staging = two weight panels");
    static_assert(NW == 4 || NW == 8, "4 or 8 waves");
    static_assert(NREQ <= 15, "vmcnt immediate");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    char* const Es = smem;
    char* const Bs = smem + NE * E_SLOT;               // panels 2, 3 double as the epilogue staging
    char* const As = Bs + NB * B_SLOT;
    char* const Scratch = As + NA * A_SLOT;            // 1 KB: where the spare requests land

    const int tid = threadIdx.x;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int wm = wave_u / WAVES_N, wn = wave_u - wm * WAVES_N;
    const int lr = lane & 31, lh = lane >> 5;
    const int pwn = (p.W + PW - 1) / PW, phn = (p.H + PH - 1) / PH;
    int nst = p.Cin / KT;                              // K steps of the current tile (a share of them when K is split)
    int ks = 0;                                        // ... and which share
    const int b_adv = p.Npad * 64;                     // F16X3: bytes between the weight panels of two K steps

    const uavsal_tile_walk walk = xcd_tile_walk(blockIdx.x, gridDim.x, p.nblk);
    int tile = walk.tile;
    if (tile >= walk.end) return;

    int img = 0, y0 = 0, x0 = 0, n0 = 0;
    // request r of this lane: running source pointer (64 bytes further every K step; lanes with nothing to fetch
    // walk a row of zeros), LDS target relative to the slot (or -1: the scratch KB)
    const char* rq_ptr[NREQ];
    int rq_lds[NREQ];
    // wave roles (a 4-wave workgroup: both; with producer waves: waves NW .. NW + 3 have both, the others neither)
    const bool loader = PWV ? wave_u >= NW : wave_u >= NW - NLW, dw_wave = PWV ? wave_u >= NW : wave_u < 4;
    const int lw = PWV ? wave_u - NW : wave_u - (NW - NLW);
#pragma unroll
    for (int r = 0; r < NREQ; ++r) {
        const int q = lw + (r < E_IT ? r : r - E_IT) * NLW;
        rq_lds[r] = (r < E_IT ? q <= E_REQ : q < B_REQ) ? q * 1024 : -1;
    }
    auto setup_tile = [&](int t) {
        int s0 = 0;                                    // first K step of this workgroup's share
        if (p.ksplit > 1) {                            // (narrow outputs: tiles_n == 1) t = tile * ksplit + share
            const int nall = p.Cin / KT;
            ks = t % p.ksplit;
            t /= p.ksplit;
            s0 = ks * nall / p.ksplit;
            nst = (ks + 1) * nall / p.ksplit - s0;
        }
        const int tm = t / p.tiles_n;
        n0 = (t - tm * p.tiles_n) * BN;
        img = tm / (phn * pwn);
        const int rem = tm - img * (phn * pwn);
        const int pyi = rem / pwn;
        y0 = pyi * PH;
        x0 = (rem - pyi * pwn) * PW;
        const int ck = (lane & 3) * 4;
        if (!loader) return;
#pragma unroll
        for (int r = 0; r < NREQ; ++r) {
            const float* src = nullptr;
            if (r < E_IT) {
                const int q = lw + r * NLW;
                if (q < E_REQ) {
                    const int hs = q * 16 + (lane >> 2);
                    const int hy = hs / HPITCH, hx = hs - hy * HPITCH;
                    const int y = y0 - 1 + hy, x = x0 - 1 + hx;
                    if (hs < NHSLOT && hx < PW + 2 && y >= 0 && y < p.H && x >= 0 && x < p.W)
                        src = p.a + ((long long)img * p.a_is + (long long)y * p.W + x) * p.lda + ck;
                } else if (q == E_REQ) {
                    const int seg = lane >> 2;
                    if (seg < 9) src = p.dw_w + (size_t)seg * p.Cin + ck;
                    else if (seg == 9) src = p.dw_s + ck;
                    else if (seg == 10) src = p.dw_b + ck;
                }
            } else {
                const int q = lw + (r - E_IT) * NLW;
                const int row = q * 16 + (lane >> 2);
                const int lc = (lane & 3) ^ ((row >> 2) & 3);
                // rows past Npad (and the spare requests) fetch the last real row: a lane of a weight request must never
                // walk the zero row, its per-step advance is a whole panel (F16X3); those columns are never stored
                const int nn = min(n0 + row, p.Npad - 1);
                src = H16 ? reinterpret_cast<const float*>(p.w + (size_t)nn * 64) + lc * 4       // [step][Npad][64 B]
                          : reinterpret_cast<const float*>(p.w) + (size_t)nn * p.Kpad + lc * 4;
            }
            rq_ptr[r] = reinterpret_cast<const char*>(src ? src : g_zero_row + ck) +
                        (size_t)s0 * ((H16 && r >= E_IT) ? b_adv : KT * 4);
        }
    };
#ifdef UAVSAL_PROBE      // tools/dwproj_probe.py parts: act = 128 + bits {1 no MFMAs, 2 no depthwise, 4 no DMA requests,
                         // 8 no fragment loads, 16 no barriers}
    const int pr_bits = p.act >= 128 ? p.act - 128 : 0;
    const bool pr_mul = !(pr_bits & 1), pr_dw = !(pr_bits & 2), pr_dma = !(pr_bits & 4), pr_frag = !(pr_bits & 8),
               pr_bar = !(pr_bits & 16);
#else
    constexpr bool pr_mul = true, pr_dw = true, pr_dma = true, pr_frag = true, pr_bar = true;
#endif
    // request r (compile-time) of the next K step to be requested: E slot at byte offset eo, weight panel at bo
    auto issue_one = [&](int r, int eo, int bo) {
        if (!pr_dma) return;
        char* dst = rq_lds[r] < 0 ? Scratch : (r < E_IT ? Es + eo : Bs + bo) + rq_lds[r];
        __builtin_amdgcn_global_load_lds((gptr_t)rq_ptr[r], (lptr_t)dst, 16, 0, 0);
        rq_ptr[r] += (H16 && r >= E_IT) ? b_adv : KT * 4;
    };
    auto issue_all = [&](int eo, int bo) {
        if (!loader) return;
#pragma unroll
        for (int r = 0; r < NREQ; ++r) issue_one(r, eo, bo);
    };
    // depthwise of one K step: E slot at es -> A tile at
    auto depthwise = [&](const char* es, char* at) {
        __builtin_amdgcn_s_setprio(1);         // the wave's vector work ahead of its SIMD-mate's (the mate only issues MFMAs)
        const int w4 = wave_u & 3;
        const int cq = lane & 3, sx2 = (lane >> 2) & 3, syl = lane >> 4, half = w4 & 1, xh = w4 >> 1;
        const char* eb = es + ((4 * half + syl) * HPITCH + 8 * xh + 2 * sx2) * 64 + cq * 16;
        const char* wb = es + W_OFF + cq * 16;
        f32x4 o[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) o[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            f32x4 e[4], w[3];
#pragma unroll
            for (int j = 0; j < 4; ++j) e[j] = *reinterpret_cast<const f32x4*>(eb + (dy * HPITCH + j) * 64);
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) w[dx] = *reinterpret_cast<const f32x4*>(wb + (dy * 3 + dx) * 64);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    o[j].x = fmaf(e[j + dx].x, w[dx].x, o[j].x); o[j].y = fmaf(e[j + dx].y, w[dx].y, o[j].y);
                    o[j].z = fmaf(e[j + dx].z, w[dx].z, o[j].z); o[j].w = fmaf(e[j + dx].w, w[dx].w, o[j].w);
                }
        }
        const f32x4 sc = *reinterpret_cast<const f32x4*>(wb + 9 * 64);
        const f32x4 bi = *reinterpret_cast<const f32x4*>(wb + 10 * 64);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f32x4 d;
            d.x = __builtin_amdgcn_fmed3f(fmaf(o[j].x, sc.x, bi.x), 0.f, 6.f);
            d.y = __builtin_amdgcn_fmed3f(fmaf(o[j].y, sc.y, bi.y), 0.f, 6.f);
            d.z = __builtin_amdgcn_fmed3f(fmaf(o[j].z, sc.z, bi.z), 0.f, 6.f);
            d.w = __builtin_amdgcn_fmed3f(fmaf(o[j].w, sc.w, bi.w), 0.f, 6.f);
            const int rho = (sx2 & 1) + 2 * syl + 8 * j + 16 * (sx2 >> 1) + 32 * half + 64 * xh;
            if (H16) {       // row = [hi k 0-7 | hi k 8-15 | lo k 0-7 | lo k 8-15], 16-byte chunks swizzled like the fp32 row
                u32x2 hi, lo;
                uavsal_split4_f16(d, hi, lo);
                const int sw = (rho >> 2) & 3;
                *reinterpret_cast<u32x2*>(at + (rho * 4 + ((cq >> 1) ^ sw)) * 16 + (cq & 1) * 8) = hi;
                *reinterpret_cast<u32x2*>(at + (rho * 4 + ((2 + (cq >> 1)) ^ sw)) * 16 + (cq & 1) * 8) = lo;
            } else {
                *reinterpret_cast<f32x4*>(at + (rho * 4 + (cq ^ ((rho >> 2) & 3))) * 16) = d;
            }
        }
        __builtin_amdgcn_s_setprio(0);
    };
    // which waves run the depthwise of K step s: all four of a 4-wave workgroup, else waves 0-3 / 4-7 in turn
    // the whole tile walk, instantiated per wave role (depthwise / requests / both): the two roles share no
    // registers beyond the accumulators and fragments
    auto run = [&](auto role_dw, auto role_ld, auto role_mma) {
    constexpr bool MMA = decltype(role_mma)::value;
    f32x16 acc[WM][WN];
    f32x4 af[2][WM], bfr[2][WN];                       // fragments of the K step about to be multiplied
    auto load_frag = [&](const char* at, const char* bt, int u) {
        if (!pr_frag || !MMA) return;
        const int chunk = 2 * u + lh;
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            const int row = (wm * WM + i) * 32 + lr;
            af[u][i] = *reinterpret_cast<const f32x4*>(at + (row * 4 + (chunk ^ ((row >> 2) & 3))) * 16);
        }
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int row = (wn * WN + j) * 32 + lr;
            bfr[u][j] = *reinterpret_cast<const f32x4*>(bt + (row * 4 + (chunk ^ ((row >> 2) & 3))) * 16);
        }
    };
    // half u of the K step's MFMAs (F16X3: u = 0 is the whole step: fragment set 0 holds the hi, set 1 the lo halves);
    // a loader wave's requests for the step four ahead are slotted behind the groups
    auto multiply_half = [&](auto with_req, int u, int eo, int bo) {
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                if (pr_mul) {
                    if (H16) {
                        const f16x8 ah = __builtin_bit_cast(f16x8, af[0][i]), al = __builtin_bit_cast(f16x8, af[1][i]);
                        const f16x8 bh = __builtin_bit_cast(f16x8, bfr[0][j]), bl = __builtin_bit_cast(f16x8, bfr[1][j]);
                        if (TEPI) {      // transposed: rows = output channels, columns = pixels (same products, same order)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, al, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, ah, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, ah, acc[i][j], 0, 0, 0);
                        } else {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[i][j], 0, 0, 0);
                        }
                    } else {
                        const f32x4 av = TEPI ? bfr[u][j] : af[u][i], bv = TEPI ? af[u][i] : bfr[u][j];
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[i][j], 0, 0, 0);
                    }
                }
                if (decltype(with_req)::value) {
                    const int grp = (u * WM + i) * WN + j;
#pragma unroll
                    for (int r = grp * RPG; r < (grp + 1) * RPG && r < NREQ; ++r) issue_one(r, eo, bo);
                }
            }
    };
    // one K step of the main loop.  eo_* / ao_* / bo_*: byte offsets of the slots of this step's roles
    //   requests: step kt + 4 -> E slot eo_req, panel bo_req      depthwise: step kt + 2, E slot eo_dw -> A tile ao_dw
    //   fragments: step kt + 1 <- A tile ao_frag, panel bo_frag
    // FULL: steps kt + 1 .. kt + 4 all exist (no conditions in the body)
    auto k_step = [&](auto full, auto role_dw, auto role_ld, int kt, int eo_req, int bo_req, int eo_dw, int ao_dw,
                      int ao_frag, int bo_frag) {
        constexpr bool FULL = decltype(full)::value, DW = decltype(role_dw)::value, LD = decltype(role_ld)::value;
        // own requests of step kt + 2 landed (the youngest step may stay in flight); own LDS accesses done except the
        // WM + WN fragment reads issued after the last MFMA; then everyone's
        if (LD) {
            if (FULL || kt + 3 < nst) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NREQ) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (pr_bar) {
            if (MMA && (FULL || kt + 1 < nst)) asm volatile("s_waitcnt lgkmcnt(%0)\n\ts_barrier" :: "n"(NTAIL) : "memory");
            else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // (a producer's last LDS accesses are its A-tile writes)
        }
        if (DW && pr_dw && (FULL || kt + 2 < nst)) depthwise(Es + eo_dw, As + ao_dw);
        if (!MMA) {                  // a producer wave: the requests of step kt + 4, back to back
            if (LD && (FULL || kt + DIST < nst)) {
#pragma unroll
                for (int r = 0; r < NREQ; ++r) issue_one(r, eo_req, bo_req);
            }
            return;
        }
        if (LD && (FULL || kt + DIST < nst)) {
            multiply_half(std::true_type{}, 0, eo_req, bo_req);
            if (H16 ? false : (FULL || kt + 1 < nst)) load_frag(As + ao_frag, Bs + bo_frag, 0);
            if (!H16) multiply_half(std::true_type{}, 1, eo_req, bo_req);
        } else {
            multiply_half(std::false_type{}, 0, 0, 0);
            if (H16 ? false : (FULL || kt + 1 < nst)) load_frag(As + ao_frag, Bs + bo_frag, 0);
            if (!H16) multiply_half(std::false_type{}, 1, 0, 0);
        }
        if (H16 && (FULL || kt + 1 < nst)) load_frag(As + ao_frag, Bs + bo_frag, 0);
        if (FULL || kt + 1 < nst) load_frag(As + ao_frag, Bs + bo_frag, 1);
    };

    setup_tile(tile);
    issue_all(0, 0);
    if (nst > 1) issue_all(E_SLOT, B_SLOT);
    while (true) {
        const int cimg = img, cy0 = y0, cx0 = x0, cn0 = n0, cks = ks;
        if (MMA) {
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int g = 0; g < 16; ++g) acc[i][j][g] = 0.f;
        }
        // ---- two lead-in steps: depthwise(0), depthwise(1), fragments of step 0
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");      // steps 0, 1 landed
        if (2 < nst) issue_all(2 * E_SLOT, 2 * B_SLOT);
        if (dw_wave && pr_dw) depthwise(Es, As);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                // A tile 0 written
        if (3 < nst) issue_all(0, 3 * B_SLOT);
        if (1 < nst && dw_wave && pr_dw) depthwise(Es + E_SLOT, As + A_SLOT);
        load_frag(As, Bs, 0);
        load_frag(As, Bs, 1);
        // rotating slot offsets: e[i] / a[i] = slot of step kt + i (mod 3), b[i] = panel of step kt + i (mod 5)
        int e0 = 0, e1 = E_SLOT, e2 = 2 * E_SLOT;
        int a0 = 0, a1 = A_SLOT, a2 = NA == 3 ? 2 * A_SLOT : 0;
        int b0 = 0, b1 = B_SLOT, b2 = 2 * B_SLOT, b3 = 3 * B_SLOT, b4 = 4 * B_SLOT;
        int kt = 0;
        auto rotate = [&]() {
            int t = e0; e0 = e1; e1 = e2; e2 = t;
            if (NA == 3) { t = a0; a0 = a1; a1 = a2; a2 = t; } else { t = a0; a0 = a1; a1 = t; a2 = a0; }
            t = b0; b0 = b1; b1 = b2; b2 = b3; b3 = b4; b4 = t;
            ++kt;
        };
        // (requests of step kt + 4: E slot (kt + 1) % 3, panel (kt + 4) % 5).  One role per wave: no per-request branches
        auto k_loop = [&](auto role_dw, auto role_ld) {
            for (; kt + DIST < nst; rotate()) k_step(std::true_type{}, role_dw, role_ld, kt, e1, b4, e2, a2, a1, b1);
            for (; kt < nst; rotate()) k_step(std::false_type{}, role_dw, role_ld, kt, e1, b4, e2, a2, a1, b1);
        };
        k_loop(role_dw, role_ld);
        const bool has_next = tile + walk.stride < walk.end;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // every wave is past its LDS reads
        if (has_next) {
            tile += walk.stride;
            setup_tile(tile);
            issue_all(0, 0);                       // E slots 0, 1 / weight panels 0, 1; the staging below uses panels 2, 3
            if (nst > 1) issue_all(E_SLOT, B_SLOT);
        }
        // ---- epilogue.  TEPI (opt-in, see the macro): the MFMA operands are swapped, so the accumulator lane holds 4 x 4
        // CONSECUTIVE OUTPUT CHANNELS of one pixel (rows = channels (g & 3) + 8 (g >> 2) + 4 lh, column = pixel lr): BN /
        // activation / residual and 16-byte stores straight from registers -- no LDS staging, no barriers.
        if (TEPI) {
            if (MMA) {
                const bool part = p.ksplit > 1;
                const bool vec = !(p.ldc & 3) && !(p.Cout & 3) && !((size_t)p.out & 15) &&
                                 (!p.res || (!(p.ldr & 3) && !((size_t)p.res & 15)));
                const int act = part ? UAVSAL_ACT_NONE : p.act;
#pragma unroll
                for (int i = 0; i < WM; ++i) {
                    const int rho = (wm * WM + i) * 32 + lr;            // A-tile row -> pixel (header comment)
                    const int y = cy0 + 4 * ((rho >> 5) & 1) + ((rho >> 1) & 3);
                    const int x = cx0 + 8 * (rho >> 6) + 2 * ((rho & 1) + 2 * ((rho >> 4) & 1)) + ((rho >> 3) & 1);
                    const bool okp = y < p.H && x < p.W;
                    const long long pix = (long long)y * p.W + x;
#pragma unroll
                    for (int j = 0; j < WN; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int gn = cn0 + (wn * WN + j) * 32 + 4 * lh + 8 * q;
                            f32x4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                            if (part) {
                                if (okp && gn < p.Npad)
                                    *reinterpret_cast<f32x4*>(p.kpart + ((size_t)cks * p.M + (size_t)cimg * p.HW + (size_t)pix) * p.Npad + gn) = v;
                                continue;
                            }
                            if (!okp || gn >= p.Cout) continue;
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                const bool okn = p.scale != nullptr && gn + c < p.Cout;
                                const float sc = (okn ? p.scale[gn + c] : 1.f) * (H16 ? F16X3_ACC_SCALE : 1.f);
                                v[c] = apply_act(fmaf(v[c], sc, okn ? p.bias[gn + c] : 0.f), act);
                            }
                            float* o = p.out + ((long long)cimg * p.o_is + pix) * p.ldc + gn;
                            const float* rs = p.res ? p.res + ((long long)cimg * p.r_is + pix) * p.ldr + gn : nullptr;
                            if (vec) {
                                if (rs) v += *reinterpret_cast<const f32x4*>(rs);
                                *reinterpret_cast<f32x4*>(o) = v;
                                if (H16 && p.out_sp)     // split shadow for the GEMM that consumes this output
                                    uavsal_store_split4(p.out_sp + ((long long)cimg * p.o_is + pix) * p.ldos, gn, v);
                            } else {
#pragma unroll
                                for (int c = 0; c < 4; ++c)
                                    if (gn + c < p.Cout) o[c] = v[c] + (rs ? rs[c] : 0.f);
                            }
                        }
                }
            }
        } else {
            float* stg = reinterpret_cast<float*>(Bs + 2 * B_SLOT);
            const bool part = p.ksplit > 1;        // K split: raw partial sums out, dwproj_reduce_kernel does the rest
            const bool vec = !(p.ldc & 3) && !(p.Cout & 3) && !((size_t)p.out & 15) &&
                             (!p.res || (!(p.ldr & 3) && !((size_t)p.res & 15)));
            float sc[WN], bi[WN];
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const int c = cn0 + (wn * WN + j) * 32 + lr;
                const bool okn = p.scale != nullptr && c < p.Cout && !part;
                sc[j] = part ? 1.f : (okn ? p.scale[c] : 1.f) * (H16 ? F16X3_ACC_SCALE : 1.f);
                bi[j] = okn ? p.bias[c] : 0.f;
            }
            const int act = part ? UAVSAL_ACT_NONE : p.act;
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int w = 0; w < WAVES_M; ++w) {
                    const int pp = w * WM + i;
                    if (MMA && wm == w) {
#pragma unroll
                        for (int g = 0; g < 16; ++g) {
                            const int r = (g & 3) + 8 * (g >> 2) + 4 * lh;
#pragma unroll
                            for (int j = 0; j < WN; ++j)
                                stg[r * BN + (wn * WN + j) * 32 + lr] = apply_act(fmaf(acc[i][j][g], sc[j], bi[j]), act);
                        }
                    }
                    __syncthreads();
#pragma unroll
                    for (int it = 0; it < (32 * BN / 4 + NT - 1) / NT; ++it) {
                        const int idx = tid + it * NT;
                        const int row = idx / (BN / 4), c4 = idx - row * (BN / 4);
                        const int rho = pp * 32 + row;              // A-tile row -> pixel (header comment)
                        const int y = cy0 + 4 * ((rho >> 5) & 1) + ((rho >> 1) & 3);
                        const int x = cx0 + 8 * (rho >> 6) + 2 * ((rho & 1) + 2 * ((rho >> 4) & 1)) + ((rho >> 3) & 1);
                        const int gn = cn0 + c4 * 4;
                        if (part) {
                            if (row < 32 && y < p.H && x < p.W && gn < p.Npad)
                                *reinterpret_cast<f32x4*>(p.kpart + ((size_t)cks * p.M + (size_t)cimg * p.HW + (size_t)y * p.W + x) * p.Npad + gn) =
                                    *reinterpret_cast<const f32x4*>(stg + row * BN + c4 * 4);
                            continue;
                        }
                        if (row < 32 && y < p.H && x < p.W && gn < p.Cout) {
                            f32x4 v = *reinterpret_cast<const f32x4*>(stg + row * BN + c4 * 4);
                            const long long pix = (long long)y * p.W + x;
                            float* o = p.out + ((long long)cimg * p.o_is + pix) * p.ldc + gn;
                            const float* rs = p.res ? p.res + ((long long)cimg * p.r_is + pix) * p.ldr + gn : nullptr;
                            if (vec) {
                                if (rs) v += *reinterpret_cast<const f32x4*>(rs);
                                *reinterpret_cast<f32x4*>(o) = v;
                                if (H16 && p.out_sp)     // split shadow for the GEMM that consumes this output
                                    uavsal_store_split4(p.out_sp + ((long long)cimg * p.o_is + pix) * p.ldos, gn, v);
                            } else {
#pragma unroll
                                for (int c = 0; c < 4; ++c)
                                    if (gn + c < p.Cout) o[c] = v[c] + (rs ? rs[c] : 0.f);
                            }
                        }
                    }
                    __syncthreads();
                }
        }
        if (!has_next) break;
    }
    };
    if (PWV) {
        if (dw_wave) run(std::true_type{}, std::true_type{}, std::false_type{});
        else run(std::false_type{}, std::false_type{}, std::true_type{});
    } else if (NW == 4) run(std::true_type{}, std::true_type{}, std::true_type{});
    else if (dw_wave) run(std::true_type{}, std::false_type{}, std::true_type{});
    else run(std::false_type{}, std::true_type{}, std::true_type{});
}


template <int PREC, int WAVES_M, int WAVES_N, int WM, int WN, int PWV = 0>
int launch_dwproj_variant(const ConvK& k0, hipStream_t stream) {
    constexpr int BN = WAVES_N * WN * 32, NT = (WAVES_M * WAVES_N + PWV) * 64;
    constexpr int SMEM = 3 * (13 * 1024) + 5 * BN * 64 + 3 * 128 * 64 + 1024;   // E slots, weight panels, A tiles, scratch
    ConvK k = k0;
    k.tiles_n = (k.Cout + BN - 1) / BN;
    k.nblk = (k.M / k.HW) * ((k.H + 7) / 8) * ((k.W + 15) / 16) * k.tiles_n;
    UAVSAL_LDS_OPTIN((dwproj_kernel<PREC, WAVES_M, WAVES_N, WM, WN, PWV>), SMEM);
    const int cap = UAVSAL_PER_DEVICE(resident_grid(dwproj_kernel<PREC, WAVES_M, WAVES_N, WM, WN, PWV>, SMEM, NT));
    // The narrowest outputs (Cout <= 32: the 1536 -> 1 decoder projection) are bound by the serial depthwise /
    // request segments of a K step, not by the matrix pipe, and their 74 KB ring leaves room for two workgroups per
    // CU: when the tiles alone would leave resident slots empty, K is split over 2-4 workgroups per tile (raw partial
    // sums into the caller's workspace, summed in a fixed order by dwproj_reduce_kernel): 96 -> 72 us fp32, 88 -> 64
    // f16x3 for 8 x 45 x 80 x 1536 -> 1.  The 64-wide instance gains nothing (81 vs 77 us: one workgroup per CU by
    // registers in fp32, and 7 MB of partial sums to re-read), so it is not split.
    k.ksplit = 1;
    if (BN <= 32 && k.kpart) {
        int ksp = UAVSAL_SPLIT_REF_SLOTS / (k.nblk > 0 ? k.nblk : 1);
        if (ksp > 4) ksp = 4;
        while (ksp > 1 && (k.Cin / 16) / ksp < 12) --ksp;
        if (ksp > 1 && (long long)ksp * k.M * k.Npad * 4 <= k.kpart_bytes) k.ksplit = ksp;
    }
    k.nblk *= k.ksplit;
    const int grid = k.nblk < cap ? k.nblk : cap;
    hipLaunchKernelGGL((dwproj_kernel<PREC, WAVES_M, WAVES_N, WM, WN, PWV>), dim3(grid), dim3(NT), SMEM, stream, k);
    if (k.ksplit > 1) return launch_splitk_reduce(k, PREC == UAVSAL_PREC_F16X3 ? F16X3_ACC_SCALE : 1.0f, stream);
    return uavsal_launch_status();
}

template <int PREC>
int launch_dwproj(const ConvK& k, hipStream_t stream) {
    static const int pw_mode = [] { const char* e = getenv("UAVSAL_DWPROJ_PW"); return e ? atoi(e) : 0; }();
    if (k.Cout > 128 && ((pw_mode & 1) && PREC == UAVSAL_PREC_F32 || (pw_mode & 2) && PREC == UAVSAL_PREC_F16X3))
        return launch_dwproj_variant<PREC, 2, 4, 2, 2, 4>(k, stream);   // + 4 producer waves (1: fp32, 2: split-fp16, 3: both)
    if (k.Cout > 128) return launch_dwproj_variant<PREC, 2, 4, 2, 2>(k, stream);   // 128 x 256, 8 waves
    if (k.Cout > 64) return launch_dwproj_variant<PREC, 2, 4, 2, 1>(k, stream);    // 128 x 128, 8 waves
    if (k.Cout > 32) return launch_dwproj_variant<PREC, 4, 2, 1, 1>(k, stream);    // 128 x 64,  8 waves
    return launch_dwproj_variant<PREC, 4, 1, 1, 1>(k, stream);                     // 128 x 32,  4 waves
}


}  // namespace

int uavsal_launch_dwproj(const uavsal_gemm::ConvK& k, int prec, hipStream_t stream) {
    return prec == UAVSAL_PREC_F32 ? launch_dwproj<UAVSAL_PREC_F32>(k, stream) : launch_dwproj<UAVSAL_PREC_F16X3>(k, stream);
}

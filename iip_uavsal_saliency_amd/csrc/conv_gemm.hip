// Dense 1x1 / 3x3 convolution as an implicit GEMM on the CDNA4 matrix cores.
//
//   out[m, n] = epilogue( sum_{tap, ci} A[pixel(m) + tap offset, ci] * W[n, tap, ci] )
//   m = output pixel (image-major, NHWC), n = output channel.
//
// Replaces BasicConv2d k=1/k=3 and the pw-linear conv+BN of dwBlock (reference
// model.py:65-72, 94-95, 100-101) and ConvTWACell.forward (model_convlstm.py:276-292);
// see include/uavsal_hip.h for the contract.
//
// Structure (one 256-thread workgroup = 4 wave64; 8 waves for the split-16-bit 128 x 256 tile):
//   * block tile BM x BN, K tile KT (16 fp32 or 32 bf16 elements = one 64-byte row of a
//     panel).  A "panel" is rows x 64 B in LDS; a row holds four 16-byte chunks, chunk c
//     stored at slot c ^ ((row >> 2) & 3) so that the ds_read_b128 fragment reads of 32
//     different rows are bank-conflict free (guide: LDS XOR swizzle, T2).
//   * global -> registers -> LDS staging, double buffered: K step s+D's global loads are
//     issued before step s's MFMAs and step s+1 is written to the other LDS stage after them
//     (issue-early / write-late, one barrier per K step; the loader is one unconditional,
//     branch-free sequence across output tiles so that the vmcnt waits stay counted).
//   * fp32 activations are converted while staging: F32 keeps them, BF16 rounds to bf16,
//     BF16X3 splits x = hi + lo (two bf16) and issues hi*hi + hi*lo + lo*hi, i.e. ~16
//     mantissa bits at 3/16 of the fp32-MFMA cost.
//   * per K sub-step each lane reads ONE 16-byte chunk per 32-row fragment: for the
//     bf16 MFMA (32x32x16) that is its 8 k-values; for the fp32 MFMA (32x32x2) the 4
//     floats feed 4 consecutive MFMAs (lanes 0-31 take chunk 2s, lanes 32-63 chunk
//     2s+1 -- any k permutation is legal as long as A and B agree).
//   * epilogue straight from the accumulators (C/D layout: col = lane & 31,
//     row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)): folded BN, ReLU6 / sigmoid,
//     residual add, channel-slice store, or the ConvTWA gate + convex state update.
//   * blockIdx is remapped so that each XCD (private L2) owns a contiguous range of
//     (M tile, N tile) pairs: the N tiles that re-read one activation tile run on the
//     same L2.
#include <stdlib.h>
#include <type_traits>
#include "common.h"

#include "conv_gemm_common.h"

namespace {

// FUSE = true: the A operand is produced on the fly as relu6(bn(depthwise3x3(E))) from the expanded
// tensor E (the `D` tensor of an inverted-residual block never exists in HBM): TAPS must be 1.
template <int PREC, int WAVES_M, int WAVES_N, int WM, int WN, int TAPS, bool FUSE = false>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64,
                             (WM * WN >= 4) ? ((PREC == UAVSAL_PREC_BF16X3 || PREC == UAVSAL_PREC_F16X3 || WAVES_M * WAVES_N == 8) ? 2 : 3) : 4)
void conv_gemm_kernel(const ConvK p) {
    constexpr int NT = WAVES_M * WAVES_N * 64;     // 4 waves, or 8 for the 128 x 256 tile
    constexpr int NS = NT;                         // all of them stage operands
    static_assert(!FUSE || TAPS == 1, "the fused depthwise producer feeds a 1x1 projection");
    constexpr int BM = WAVES_M * WM * 32;
    constexpr int BN = WAVES_N * WN * 32;
    constexpr int KT = (PREC == UAVSAL_PREC_F32) ? 16 : 32;
    constexpr int NPAN = (PREC == UAVSAL_PREC_BF16X3 || PREC == UAVSAL_PREC_F16X3) ? 2 : 1;
    constexpr int NLD = (PREC == UAVSAL_PREC_F32) ? 1 : 2;   // float4 loads per A chunk
    constexpr int A_IT = (BM * 4) / NS;
    constexpr int B_IT = (BN * 4 + NS - 1) / NS;
    constexpr int APAN = BM * 64;
    constexpr int BPAN = BN * 64;
    constexpr int STAGE = NPAN * (APAN + BPAN);
    static_assert(NS == 256 || NS == 512, "4 or 8 staging waves per workgroup");
    static_assert((BM * 4) % NS == 0, "A tile must divide over the staging threads");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    // Persistent workgroups: the grid is sized to the resident capacity of the chip; every XCD owns a
    // contiguous range of output tiles (N fastest) and its workgroups take them round-robin
    // (common.h: xcd_tile_walk).  The loader below runs ahead across output-tile boundaries, which
    // hides the prologue's HBM/L2 latency -- with K as short as 256 that latency was a third of the
    // kernel's time.
    const uavsal_tile_walk walk = xcd_tile_walk(blockIdx.x, gridDim.x, p.nblk);
    int tile = walk.tile;
    const int tile_end = walk.end, tile_step = walk.stride;
    if (tile >= tile_end) return;
    int m0 = 0, n0 = 0;
    // K split (host: launch_variant): only the 64 x 64 tile carries it -- the ConvTWA step is 228 tiles with 72 K
    // steps each (the 12x20 ASPP projections: 120 tiles x 60 steps), one workgroup per CU and a serial load ->
    // convert -> multiply chain per step; `tile` then counts (output tile, share) pairs and a share's raw sums go to
    // p.kpart
    constexpr bool CAN_SPLIT = WM * WN == 1 && !FUSE;
    const int KS = CAN_SPLIT ? p.ksplit : 1;
    const int kshare = p.ktiles / KS;              // K steps per share (host: divisible, a multiple of 9 steps)
    int l_base = 0;                                // first K step of the share the loader is in

    // ---- per-thread staging coordinates ------------------------------------------------
    const int stid = tid;
    const int ch = stid & 3;             // 16-byte chunk within the 64-byte panel row
    constexpr int ESZ = (PREC == UAVSAL_PREC_F32) ? 4 : 2;
    const int b_row0 = stid >> 2;
    long long a_base[A_IT];
    int a_y[A_IT], a_x[A_IT];            // fused producer only
    int a_taps[A_IT];                    // 3x3: bit t set = tap t of this row lies inside the image
    bool a_ok[A_IT];
    unsigned b_off[B_IT];
    auto setup_tile = [&](int t) {
        if (CAN_SPLIT && KS > 1) {
            const int tt = t / KS;
            l_base = (t - tt * KS) * kshare;
            t = tt;
        }
        const int tile_m = t / p.tiles_n;
        const int tile_n = t - tile_m * p.tiles_n;
        m0 = tile_m * BM;
        n0 = tile_n * BN;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int row = (stid >> 2) + it * (NS / 4);
            const int m = m0 + row;
            a_ok[it] = m < p.M;
            const int mm = a_ok[it] ? m : 0;
            a_taps[it] = 0;
            if (TAPS == 1 && !FUSE) {
                a_base[it] = row_off(mm, p.HW, p.a_is, p.contig) * p.lda;
                a_y[it] = 0; a_x[it] = 0;
            } else if (TAPS == 9) {
                // element offset of the centre pixel + a mask of the taps inside the image: per K step
                // only a wave-uniform tap delta is added and one bit tested
                const int img = mm / p.HW;
                const int pix = mm - img * p.HW;
                const int y = pix / p.W, x = pix - y * p.W;
                a_base[it] = ((long long)img * p.a_is + pix) * p.lda;
                a_y[it] = 0; a_x[it] = 0;
                int mask = 0;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                    if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) mask |= 1 << t;
                }
                a_taps[it] = mask;
            } else {
                const int img = mm / p.HW;
                const int pix = mm - img * p.HW;
                a_y[it] = pix / p.W;
                a_x[it] = pix - a_y[it] * p.W;
                a_base[it] = (long long)img * p.a_is;   // in pixels; tap offset added per tile
            }
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            // rows past the padded weight matrix are clamped, not zeroed: they only feed output
            // columns >= Cout, which the epilogue never stores (and a clamp keeps the load branch-free)
            const int nn = min(n0 + b_row0 + it * (NS / 4), p.Npad - 1);
            // 16-bit layouts are K-step-major (one 64-byte row per n and step); fp32 is [Npad][Kpad]
            b_off[it] = (unsigned)nn * (PREC == UAVSAL_PREC_F32 ? (unsigned)p.Kpad * 4u : 64u) + ch * 16;
        }
    };
    // opaque to the optimiser: with a known global on one arm it turns `*(ok ? p : zero)` back into
    // a divergent branch around two loads
    size_t zero_page = (size_t)g_zero16;
    asm volatile("" : "+s"(zero_page));

    // D register sets: K step s+D is requested while step s is multiplied.  Measured on the path's
    // shapes (profiles/r1_gemm_probe_v3.log) the depth barely matters -- these kernels are bound by
    // the conversion VALU work and the barrier-phased MFMA / VALU alternation, not by load latency:
    // D = 1, 2, 3 are within 3% on the 4-wave tiles (1 is best and cheapest in registers); only the
    // 8-wave 128 x 256 tile, alone on its CU, gains from D = 2 (up to 9% on the 64-frame expands).
    // ... and the 64 x 64 3x3 tile of the ConvTWA steps (228 tiles: one workgroup per CU, nothing else covers a
    // load): D = 2 measured 50 vs 52 us per step and +1.1 % end to end in f16x3 at one clip (3: slower)
#ifndef UAVSAL_GEMM_PREFETCH_SMALL
#define UAVSAL_GEMM_PREFETCH_SMALL 2
#endif
    constexpr int D = (!FUSE && WAVES_M * WAVES_N == 8 && WM * WN <= 4) ? UAVSAL_GEMM_PREFETCH
                      : ((!FUSE && WM * WN == 1 && TAPS == 9) ? UAVSAL_GEMM_PREFETCH_SMALL : 1);
    f32x4 a_reg[D][A_IT][NLD];
    u32x4 b_reg[D][B_IT][NPAN];

    auto load_tile = [&](int kt, int ci0, int tap, int set) {
        long long tap_off = 0;                              // wave-uniform: (dy * W + dx) * lda
        if (TAPS == 9) {
            const int ty = (tap * 11) >> 5;                 // tap / 3 for tap in 0..9
            tap_off = (long long)((ty - 1) * p.W + (tap - ty * 3 - 1)) * p.lda;
        }
        if (FUSE) {
            // depthwise 3x3 (+BN+ReLU6) of E for this thread's rows and 4-channel groups
#pragma unroll
            for (int l = 0; l < NLD; ++l) {
                const int kk = ci0 + ch * 4 + l * 16;
                const bool kin = kk < p.Cin;
                const int kc = kin ? kk : 0;
                f32x4 accd[A_IT];
#pragma unroll
                for (int it = 0; it < A_IT; ++it) accd[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const f32x4 wv = *reinterpret_cast<const f32x4*>(p.dw_w + (size_t)(ky * 3 + kx) * p.Cin + kc);
#pragma unroll
                        for (int it = 0; it < A_IT; ++it) {
                            const int iy = a_y[it] * p.dw_stride - 1 + ky, ix = a_x[it] * p.dw_stride - 1 + kx;
                            const bool ok = a_ok[it] && kin && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
                            if (ok)
                                accd[it] += *reinterpret_cast<const f32x4*>(
                                    p.a + (a_base[it] + (long long)iy * p.Win + ix) * p.lda + kk) * wv;
                        }
                    }
                }
                const f32x4 sv = *reinterpret_cast<const f32x4*>(p.dw_s + kc);
                const f32x4 bv = *reinterpret_cast<const f32x4*>(p.dw_b + kc);
#pragma unroll
                for (int it = 0; it < A_IT; ++it) {
                    f32x4 d = accd[it] * sv + bv;
                    d.x = fminf(fmaxf(d.x, 0.f), 6.f); d.y = fminf(fmaxf(d.y, 0.f), 6.f);
                    d.z = fminf(fmaxf(d.z, 0.f), 6.f); d.w = fminf(fmaxf(d.w, 0.f), 6.f);
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    a_reg[set][it][l] = (a_ok[it] && kin) ? d : z;
                }
            }
        } else {
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            bool ok = a_ok[it];
            if (TAPS == 9) ok = ok && ((a_taps[it] >> tap) & 1);
            const long long off = a_base[it] + tap_off;
#pragma unroll
            for (int l = 0; l < NLD; ++l) {
                const int kk = ci0 + ch * 4 + l * 16;
                // out-of-range lanes read the zero page: an unconditional load keeps the loop free of
                // divergent branches, so the compiler can count vmcnt instead of draining to 0
                const bool okk = ok && kk < p.Cin;
                a_reg[set][it][l] = *(const __attribute__((address_space(1))) f32x4*)(
                    okk ? (size_t)(p.a + off + kk) : zero_page);
            }
        }
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
#pragma unroll
            for (int pn = 0; pn < NPAN; ++pn) {
                // wave-uniform base (panel, K step) + the lane's 32-bit row offset from setup_tile
                const char* base = (PREC == UAVSAL_PREC_F32) ? p.w + (size_t)kt * 64
                                                             : p.w + ((size_t)kt * NPAN + pn) * p.Npad * 64;
                b_reg[set][it][pn] = *reinterpret_cast<const u32x4*>(base + b_off[it]);
            }
        }
    };

    auto store_tile = [&](int set, int stage) {
        char* As = smem + stage * STAGE;
        char* Bs = As + NPAN * APAN;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int row = (stid >> 2) + it * (NS / 4);
            const int slot = (row * 4 + (ch ^ ((row >> 2) & 3))) * 16;
            if (PREC == UAVSAL_PREC_F32) {
                *reinterpret_cast<f32x4*>(As + slot) = a_reg[set][it][0];
            } else if (PREC == UAVSAL_PREC_F16X3) {
                // The conversion VALU work bounds this kernel, so the split is kept to ~3.5 instructions
                // per element: packed scale, v_cvt_pkrtz (two elements per instruction; round-toward-zero
                // overflows to the largest finite half, so no clamp is needed and nothing becomes inf),
                // unpack, packed subtract, v_cvt_pkrtz.  lo = x - hi is exact in fp32 whatever the
                // rounding of hi, so RTZ costs one bit of the ~21 (2^-20 relative), not correctness.
                typedef float f2 __attribute__((ext_vector_type(2)));
                typedef __fp16 h2 __attribute__((ext_vector_type(2)));
                u32x4 hi, lo;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 src = a_reg[set][it][(q >> 1) * (NLD - 1)];
                    f2 x = (q & 1) ? (f2){src.z, src.w} : (f2){src.x, src.y};
                    x = x * F16X3_A_SCALE;
                    const h2 h = __builtin_amdgcn_cvt_pkrtz(x.x, x.y);
                    const f2 r = x - (f2){(float)h.x, (float)h.y};
                    const h2 l = __builtin_amdgcn_cvt_pkrtz(r.x, r.y);
                    hi[q] = __builtin_bit_cast(unsigned, h);
                    lo[q] = __builtin_bit_cast(unsigned, l);
                }
                *reinterpret_cast<u32x4*>(As + slot) = hi;
                *reinterpret_cast<u32x4*>(As + APAN + slot) = lo;
            } else {
                const f32x4 x0 = prescale<PREC>(a_reg[set][it][0]), x1 = prescale<PREC>(a_reg[set][it][NLD - 1]);
                u32x4 hi;
                hi.x = pack2<PREC>(x0.x, x0.y); hi.y = pack2<PREC>(x0.z, x0.w);
                hi.z = pack2<PREC>(x1.x, x1.y); hi.w = pack2<PREC>(x1.z, x1.w);
                *reinterpret_cast<u32x4*>(As + slot) = hi;
                if (NPAN == 2) {
                    u32x4 lo;
                    lo.x = pack2<PREC>(x0.x - hi_as_f32<PREC>(x0.x), x0.y - hi_as_f32<PREC>(x0.y));
                    lo.y = pack2<PREC>(x0.z - hi_as_f32<PREC>(x0.z), x0.w - hi_as_f32<PREC>(x0.w));
                    lo.z = pack2<PREC>(x1.x - hi_as_f32<PREC>(x1.x), x1.y - hi_as_f32<PREC>(x1.y));
                    lo.w = pack2<PREC>(x1.z - hi_as_f32<PREC>(x1.z), x1.w - hi_as_f32<PREC>(x1.w));
                    *reinterpret_cast<u32x4*>(As + APAN + slot) = lo;
                }
            }
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const int row = b_row0 + it * (NS / 4);
            if ((BN * 4) % NS == 0 || row < BN) {
                const int slot = (row * 4 + (ch ^ ((row >> 2) & 3))) * 16;
#pragma unroll
                for (int pn = 0; pn < NPAN; ++pn)
                    *reinterpret_cast<u32x4*>(Bs + pn * BPAN + slot) = b_reg[set][it][pn];
            }
        }
    };

    f32x16 acc[WM][WN];

    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave / WAVES_N, wn = wave - wm * WAVES_N;
    const int lr = lane & 31, lh = lane >> 5;

    auto compute = [&](int stage) {
        const char* As = smem + stage * STAGE;
        const char* Bs = As + NPAN * APAN;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int chunk = 2 * s + lh;
            u32x4 af[WM][NPAN], bfr[WN][NPAN];
#pragma unroll
            for (int i = 0; i < WM; ++i) {
                const int row = (wm * WM + i) * 32 + lr;
                const int slot = (row * 4 + (chunk ^ ((row >> 2) & 3))) * 16;
#pragma unroll
                for (int pn = 0; pn < NPAN; ++pn)
                    af[i][pn] = *reinterpret_cast<const u32x4*>(As + pn * APAN + slot);
            }
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const int row = (wn * WN + j) * 32 + lr;
                const int slot = (row * 4 + (chunk ^ ((row >> 2) & 3))) * 16;
#pragma unroll
                for (int pn = 0; pn < NPAN; ++pn)
                    bfr[j][pn] = *reinterpret_cast<const u32x4*>(Bs + pn * BPAN + slot);
            }
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    if (PREC == UAVSAL_PREC_F32) {
                        const f32x4 av = __builtin_bit_cast(f32x4, af[i][0]);
                        const f32x4 bv = __builtin_bit_cast(f32x4, bfr[j][0]);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[i][j], 0, 0, 0);
                    } else if (PREC == UAVSAL_PREC_F16X3) {
                        const f16x8 ah = __builtin_bit_cast(f16x8, af[i][0]);
                        const f16x8 bh = __builtin_bit_cast(f16x8, bfr[j][0]);
                        const f16x8 al = __builtin_bit_cast(f16x8, af[i][NPAN - 1]);
                        const f16x8 bl = __builtin_bit_cast(f16x8, bfr[j][NPAN - 1]);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[i][j], 0, 0, 0);
                    } else {
                        const bf16x8 ah = __builtin_bit_cast(bf16x8, af[i][0]);
                        const bf16x8 bh = __builtin_bit_cast(bf16x8, bfr[j][0]);
                        if (PREC == UAVSAL_PREC_BF16X3) {
                            const bf16x8 al = __builtin_bit_cast(bf16x8, af[i][NPAN - 1]);
                            const bf16x8 bl = __builtin_bit_cast(bf16x8, bfr[j][NPAN - 1]);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[i][j], 0, 0, 0);
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[i][j], 0, 0, 0);
                        }
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[i][j], 0, 0, 0);
                    }
                }
        }
    };

    // ---- the loader: one flat sequence of (output tile, K step) positions ---------------------
    // It runs D positions ahead of the MFMAs and crosses output-tile boundaries, so the first K
    // tiles of the next output tile land while the current tile's epilogue runs.  Every output tile
    // takes S = roundup(ktiles, D) steps (the pad steps re-request the last K tile and multiply
    // nothing), so a tile always starts in register set 0.  EVERY step issues the same number of
    // loads, unconditionally: with a load behind a branch -- even a uniform one -- the compiler's
    // waitcnt pass merges the two paths and waits for vmcnt(0), which is prefetch distance 1 again.
    const int S = (kshare + D - 1) / D * D;
    int l_tile = tile, l_s = 0;                  // loader position
    int lt_tap = 0, lt_ci = 0;                   // its running K position (advanced in order)
    int l_kt = 0, l_ci0 = 0, l_tapv = 0;         // what the last real step requested
    setup_tile(l_tile);
    if (CAN_SPLIT) lt_ci = (TAPS == 9 ? l_base / 9 : l_base) * KT;    // (3x3: a share starts on a channel block, tap 0)
    auto loader_step = [&](int set) {
        if (l_s < kshare) {
            l_kt = l_base + l_s; l_ci0 = lt_ci; l_tapv = lt_tap;
            // 3x3: K runs channel-block-major, tap-minor -- the nine taps of one 64/128-byte channel
            // chunk are nine consecutive K steps, so every fetched line sees all its uses while it
            // is still in L2 (tap-major re-fetched the activations ~6x: PMC FETCH_SIZE)
            if (TAPS == 9) { if (++lt_tap == 9) { lt_tap = 0; lt_ci += KT; } }
            else lt_ci += KT;
        }
        load_tile(l_kt, l_ci0, l_tapv, set);
        if (++l_s == S) {
            l_s = 0; lt_tap = 0;
            if (l_tile + tile_step < tile_end) {  // past the last tile: keep re-requesting it (harmless)
                l_tile += tile_step;
                setup_tile(l_tile);
            }
            lt_ci = CAN_SPLIT ? (TAPS == 9 ? l_base / 9 : l_base) * KT : 0;
        }
    };

    // ---- persistent loop over this workgroup's output tiles -------------------------------
#pragma unroll
    for (int d = 0; d < D; ++d) loader_step(d);
    while (true) {
        const int otile = (CAN_SPLIT && KS > 1) ? tile / KS : tile;
        const int cks = tile - otile * KS;                                      // (share being computed)
        const int tile_m = otile / p.tiles_n;
        const int m0c = tile_m * BM, n0c = (otile - tile_m * p.tiles_n) * BN;  // tile being computed
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int g = 0; g < 16; ++g) acc[i][j][g] = 0.f;

        store_tile(0, 0);
        __syncthreads();
        for (int s0 = 0; s0 < S; s0 += D) {
#pragma unroll
            for (int u = 0; u < D; ++u) {          // K step s lives in register set s % D == u
                const int st = s0 + u;
                loader_step(u);                    // position st + D; set u went to LDS last step
                // keep the order request -> multiply -> convert: hoisting the conversion of step
                // st+1 above these requests would drain vmcnt first
                if (D > 1) __builtin_amdgcn_sched_barrier(0);
                if (st < kshare) compute(st & 1);
                if (D > 1) __builtin_amdgcn_sched_barrier(0);
                if (st + 1 < kshare) store_tile((u + 1) % D, (st + 1) & 1);
                __syncthreads();
            }
        }

        // ---- epilogue ------------------------------------------------------------------
        if (CAN_SPLIT && KS > 1) {
            // a share: raw sums out (32 consecutive floats per lane row), splitk_reduce_kernel does the rest
            if constexpr (CAN_SPLIT) {
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int m = m0c + wm * 32 + (g & 3) + 8 * (g >> 2) + 4 * lh;
                    const int n = n0c + wn * 32 + lr;
                    if (m < p.M && n < p.Npad) p.kpart[((size_t)cks * p.M + m) * p.Npad + n] = acc[0][0][g];
                }
            }
        } else {
            UAVSAL_GEMM_EPILOGUE((PREC == UAVSAL_PREC_F16X3 ? F16X3_ACC_SCALE : 1.0f), smem, (PREC == UAVSAL_PREC_F16X3))
        }
        tile += tile_step;
        if (tile >= tile_end) break;
    }
}

// =====================================================================================
// fp32 path: LDS-DMA staged, multi-stage ring.
//
// fp32 operands need no conversion, so both tiles go global -> LDS directly
// (global_load_lds_dwordx4: no VGPRs, no VALU, no ds_write) into a ring of S stages that is
// kept D = S-1 K tiles ahead of the MFMAs.  One raw s_barrier per K tile:
//   wait  : s_waitcnt vmcnt(LPT*(D-1))  -> this wave's DMA for tile kt has landed
//   barrier: every wave's has (RAW), and every wave has finished reading the stage tile
//            kt+D will overwrite (it was read in iteration kt-1)              (WAR)
//   issue : DMA for tile kt+D
//   compute tile kt from stage kt % S
// The LDS destination of a DMA is wave-uniform base + lane*16, so the chunk swizzle is applied
// on the per-lane SOURCE address (lane l of a 16-row group writes physical slot l&3 of row l>>2
// and therefore fetches logical chunk (l&3) ^ ((row>>2)&3)); fragment reads use the same XOR.
// Out-of-range lanes (M / K tails, 3x3 zero padding) fetch from a 16-byte zero page instead of
// being masked off (a masked lane would leave stale LDS behind).


// SK = true (stream-K): the K stages of an XCD's tiles form one line that is cut into equal
// ranges, one per workgroup, so every CU gets the same number of MFMA stages even when there are
// only 1.76 tiles per CU (one clip: 450 tiles of the K=1536 projections and of the 3x3 convs, where
// whole-tile scheduling idles 12-15 % of the chip).  A range is [tail piece of a tile][whole
// tiles][head piece of a tile].  A tail piece is computed FIRST and published (fp32 partial +
// flag, device-coherent accesses); the workgroup that owns a tile's first stage computes its head
// piece LAST, adds the published pieces in a fixed order and runs the epilogue -- nobody waits
// on work that has not been started, and the summation order is fixed by the grid.
template <int WAVES_M, int WAVES_N, int WM, int WN, int TAPS, int S, int NKP, bool SK = false>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64, WAVES_M * WAVES_N == 8 ? 4 : (SK ? 2 : ((WM * WN >= 4) ? 3 : 4)))
void conv_gemm_f32_dma_kernel(const ConvK p) {
    constexpr int BM = WAVES_M * WM * 32;
    constexpr int BN = WAVES_N * WN * 32;
    constexpr int KT = 16;                       // one 64-byte panel row; a stage holds NKP panels
    constexpr int NW = WAVES_M * WAVES_N;        // 4 waves, or 8 for the 256 x 128 tile (whole-tile path only)
    constexpr int NT = NW * 64;
    constexpr int RPI = NT / 4;                  // panel rows one DMA iteration of the workgroup covers
    constexpr int A_IT = (BM * 4) / NT;
    constexpr int B_IT = (BN * 4 + NT - 1) / NT;
    constexpr int BWR = (BN / 16 < NW) ? BN / 16 : NW;   // waves with distinct B rows in one iteration
    constexpr int LPT = NKP * (A_IT + B_IT);     // DMA instructions per thread per stage
    constexpr int D = S - 1;                     // prefetch distance in stages
    constexpr int APAN = BM * 64;
    constexpr int BPAN = BN * 64;
    constexpr int PANEL = APAN + BPAN;
    constexpr int STAGE = NKP * PANEL;
    static_assert(NW == 4 || (NW == 8 && !SK), "4 waves per workgroup (8 for the whole-tile 256 x 128 instance)");
    static_assert((BM * 4) % NT == 0, "A tile must divide over the workgroup");
    static_assert(BN == 32 || BN == 64 || BN == 128, "B tile shape");
    static_assert(LPT * (D - 1) < 64, "vmcnt immediate");
    static_assert(STAGE >= 32 * BN * 4, "epilogue staging must fit one ring stage");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

    const int tid = threadIdx.x;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably wave-uniform
    const uavsal_tile_walk walk = xcd_tile_walk(blockIdx.x, gridDim.x, p.nblk);
    int tile = walk.tile;
    const int tile_end = walk.end, tile_step = walk.stride;
    int m0 = 0, n0 = 0;

    // per-thread DMA coordinates: row (tid>>2) + it*RPI of the tile, physical chunk tid&3
    const int pc = tid & 3;
    long long a_base[A_IT];          // element offset of the row's centre pixel
    int a_taps[A_IT], a_lc[A_IT];    // 3x3: bit t set = tap t of this row lies inside the image
    bool a_ok[A_IT];
    int b_lc[B_IT];
    const float* zero = g_zero16;
    auto setup_tile = [&](int t) {
        const int tile_m = t / p.tiles_n;
        const int tile_n = t - tile_m * p.tiles_n;
        m0 = tile_m * BM;
        n0 = tile_n * BN;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int row = (tid >> 2) + it * RPI;
            a_lc[it] = pc ^ ((row >> 2) & 3);
            const int m = m0 + row;
            a_ok[it] = m < p.M;
            const int mm = a_ok[it] ? m : 0;
            if (TAPS == 1) {
                a_base[it] = row_off(mm, p.HW, p.a_is, p.contig) * p.lda;
                a_taps[it] = 0;
            } else {
                // per step only a wave-uniform tap delta is added and one mask bit tested: the DMA
                // issue path is what bounds this kernel, so no per-step bounds arithmetic
                const int img = mm / p.HW;
                const int pix = mm - img * p.HW;
                const int y = pix / p.W, x = pix - y * p.W;
                a_base[it] = ((long long)img * p.a_is + pix) * p.lda;
                int mask = 0;
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
                    if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) mask |= 1 << t;
                }
                a_taps[it] = mask;
            }
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const int wrow = ((tid >> 6) % BWR) * 16;
            const int row = wrow + ((tid & 63) >> 2) + it * RPI;
            b_lc[it] = pc ^ ((row >> 2) & 3);
        }
    };

    const int kpanels = p.Kpad / KT;             // real 16-wide K panels; stages beyond are zero
    // running K position of the next panel to request (panels are requested strictly in order,
    // so no division by Cin is needed to find the tap of a panel)
    int it_kt = 0, it_tap = 0, it_ci = 0;
    auto issue_panel = [&](int q, int stage) {
        {
            const int kt = it_kt;
            const bool kin = kt < kpanels;
            char* As = smem + stage * STAGE + q * PANEL;
            char* Bs = As + APAN;
            const int ci0 = it_ci;
            const int tap = it_tap;
            long long tap_off = 0;                          // wave-uniform: (dy * W + dx) * lda
            if (TAPS == 9) {
                const int ty = (tap * 11) >> 5;             // tap / 3 for tap in 0..9
                tap_off = (long long)((ty - 1) * p.W + (tap - ty * 3 - 1)) * p.lda;
            }
            ++it_kt;
            if (TAPS == 9) { if (++it_tap == 9) { it_tap = 0; it_ci += KT; } }   // channel block major, tap minor
            else it_ci += KT;
#pragma unroll
            for (int it = 0; it < A_IT; ++it) {
                bool ok = a_ok[it] && kin;
                if (TAPS == 9) ok = ok && ((a_taps[it] >> tap) & 1);
                const long long off = a_base[it] + tap_off;
                const int kk = ci0 + a_lc[it] * 4;
                const float* src = (ok && kk < p.Cin) ? (p.a + off + kk) : zero;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(As + (it * RPI + wave_u * 16) * 64), 16, 0, 0);
            }
#pragma unroll
            for (int it = 0; it < B_IT; ++it) {
                // BN = 32: the panel has only 32 rows; waves 2,3 re-request the rows of waves 0,1
                // (identical bytes to identical addresses) so that EVERY wave issues exactly LPT
                // requests per stage -- the counted s_waitcnt vmcnt below depends on it.
                const int wrow = (wave_u % BWR) * 16;
                const int row = wrow + ((tid & 63) >> 2) + it * RPI;
                const int nn = n0 + row;
                const float* src = zero;
                if (kin && row < BN && nn < p.Npad)
                    src = reinterpret_cast<const float*>(p.w) + (size_t)nn * p.Kpad + (size_t)kt * KT + b_lc[it] * 4;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(Bs + (it * RPI + wrow) * 64), 16, 0, 0);
            }
        }
    };
    auto issue_tile = [&](int /*st_idx*/, int stage) {
#pragma unroll
        for (int q = 0; q < NKP; ++q) issue_panel(q, stage);
    };

    f32x16 acc[WM][WN];
    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave / WAVES_N, wn = wave - wm * WAVES_N;
    const int lr = lane & 31, lh = lane >> 5;

    // Fragment reads are double buffered by hand (sub-step u+1's ds_reads are in flight during
    // sub-step u's MFMAs) and the next stage's DMA requests are slotted in after each panel: with
    // one workgroup per CU (the ConvTWA step: 228 tiles) there is no other wave on the SIMD to
    // cover LDS latency or DMA issue time.
    auto compute = [&](int stage, bool do_issue, int istage) {
        f32x4 af[2][WM], bfr[2][WN];
        auto ld = [&](int u, f32x4 (&a)[WM], f32x4 (&b)[WN]) {
            const char* As = smem + stage * STAGE + (u >> 1) * PANEL;
            const char* Bs = As + APAN;
            const int chunk = 2 * (u & 1) + lh;
#pragma unroll
            for (int i = 0; i < WM; ++i) {
                const int row = (wm * WM + i) * 32 + lr;
                a[i] = *reinterpret_cast<const f32x4*>(As + (row * 4 + (chunk ^ ((row >> 2) & 3))) * 16);
            }
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const int row = (wn * WN + j) * 32 + lr;
                b[j] = *reinterpret_cast<const f32x4*>(Bs + (row * 4 + (chunk ^ ((row >> 2) & 3))) * 16);
            }
        };
        ld(0, af[0], bfr[0]);
#pragma unroll
        for (int u = 0; u < 2 * NKP; ++u) {
            if (u + 1 < 2 * NKP) ld(u + 1, af[(u + 1) & 1], bfr[(u + 1) & 1]);
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    const f32x4 av = af[u & 1][i], bv = bfr[u & 1][j];
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[i][j], 0, 0, 0);
                }
            if ((u & 1) && do_issue) issue_panel(u >> 1, istage);
        }
    };

    const int nst = (kpanels + NKP - 1) / NKP;   // stages (groups of NKP panels) along K
    if constexpr (SK) {
        __shared__ int sk_piece_ok;
        // ---- this XCD's tiles and this workgroup's range of their K stages -------------------
        const int G = gridDim.x, q = G / UAVSAL_NUM_XCD, r = G % UAVSAL_NUM_XCD;
        const int xcd = blockIdx.x % UAVSAL_NUM_XCD, slot = blockIdx.x / UAVSAL_NUM_XCD;
        const int gx = q + (xcd < r ? 1 : 0);
        const int vb0 = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        const int T0 = (int)(((long long)vb0 * p.nblk) / G);
        const int T1 = (int)(((long long)(vb0 + gx) * p.nblk) / G);
        const long long Wx = (long long)(T1 - T0) * nst;
        auto bound = [&](int j) { return (int)((Wx * j) / gx); };     // first stage of workgroup j
        int cur = bound(slot);
        const int w1 = bound(slot + 1);
        const int lane_idx = tid;                                     // partials are stored in register order
        struct Seg { int tl, s_begin, nseg, npre, m0c, n0c; };
        // opens the segment that starts at stage `c` of the XCD's line: tile coordinates, K position,
        // and the first D stages requested
        auto open_segment = [&](int c) {
            Seg g;
            g.tl = c / nst;
            g.s_begin = c - g.tl * nst;
            g.nseg = min(nst - g.s_begin, w1 - c);
            tile = T0 + g.tl;
            setup_tile(tile);
            g.m0c = m0; g.n0c = n0;
            it_kt = g.s_begin * NKP;
            if (TAPS == 9) { it_tap = it_kt % 9; it_ci = (it_kt / 9) * KT; }
            else { it_tap = 0; it_ci = it_kt * KT; }
            g.npre = g.nseg < D ? g.nseg : D;
            for (int t = 0; t < g.npre; ++t) issue_tile(t, t);
            return g;
        };
        Seg sg = {};
        if (cur < w1) sg = open_segment(cur);
        while (cur < w1) {
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j)
#pragma unroll
                    for (int g = 0; g < 16; ++g) acc[i][j][g] = 0.f;
            int stage = 0, istage = sg.npre % S;
            for (int kt = 0; kt < sg.nseg; ++kt) {
                if (kt + D <= sg.nseg) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LPT * (D - 1)) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                const bool do_issue = kt + D < sg.nseg;
                compute(stage, do_issue, istage);
                if (do_issue) istage = (istage + 1 == S) ? 0 : istage + 1;
                stage = (stage + 1 == S) ? 0 : stage + 1;
            }
            __builtin_amdgcn_s_barrier();        // ring reads done: stages 0..D-1 take the next segment's
                                                 // first requests, stage S-1 the epilogue staging
            const int tl = sg.tl, s_begin = sg.s_begin, m0c = sg.m0c, n0c = sg.n0c;
            cur += sg.nseg;
#if UAVSAL_SK_PREFETCH
            if (cur < w1) sg = open_segment(cur);    // lands while this segment is published / finished
#endif
            if (s_begin > 0) {
                // a later piece of a tile whose first stage belongs to another workgroup: publish
                float* part = p.sk_part + (size_t)(vb0 + slot) * (BM * BN);
                // device-coherent (sc1) 16-byte stores: dword stores at agent scope move these 64 KB at a
                // fraction of the rate and put ~30 us on the critical path of a 230 us launch
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j)
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4) {
                            const f32x4 v = {acc[i][j][4 * q4], acc[i][j][4 * q4 + 1], acc[i][j][4 * q4 + 2],
                                             acc[i][j][4 * q4 + 3]};
                            float* dst = part + ((size_t)((i * WN + j) * 4 + q4) * 256 + lane_idx) * 4;
                            asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(dst), "v"(v) : "memory");
                        }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                // (sk_drop: test hook -- this workgroup "loses" its piece, tests/test_hip_kernels.py)
                if (tid == 0 && p.sk_drop != vb0 + slot + 1 && p.sk_drop >= 0)
                    __hip_atomic_store(p.sk_flag + vb0 + slot, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                // owner of the tile's first stage: add the pieces the following workgroups published.
                // Hand-off (MI355X_MICROARCH.md, inter-workgroup visibility): producer = sc1 write-through
                // stores, every wave's vmcnt(0), workgroup barrier, flag; consumer = ONE relaxed poll, ONE agent
                // acquire (this CU's L1), vmcnt(0), workgroup barrier, then the loads (kept sc1: L2-served).
                int e = cur, k = slot + 1;
                while (e < (tl + 1) * nst) {
                    if (tid == 0) {
                        int spins = 0;       // bounded: a lost piece must never hang the GPU
                        while (__hip_atomic_load(p.sk_flag + vb0 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 &&
                               ++spins < p.sk_spin)
                            __builtin_amdgcn_s_sleep(8);
                        const int ok = spins < p.sk_spin;
                        // gave up: the piece stays OUT of the sum (its buffer holds an older launch's data) and the
                        // error word makes the run invalid for the host (uavsal_guard poisons the outputs)
                        if (!ok) __hip_atomic_fetch_or(p.err, UAVSAL_ERR_STREAMK, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        sk_piece_ok = ok;
#if UAVSAL_SK_ACQUIRE
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                    }
                    __syncthreads();
                    const bool have = sk_piece_ok != 0;
                    const float* part = p.sk_part + (size_t)(vb0 + k) * (BM * BN);
                    if (have) {
#pragma unroll
                    for (int i = 0; i < WM; ++i) {
                        f32x4 t[WN * 4];
#pragma unroll
                        for (int n = 0; n < WN * 4; ++n) {
                            const float* src = part + ((size_t)(i * WN * 4 + n) * 256 + lane_idx) * 4;
                            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(t[n]) : "v"(src) : "memory");
                        }
                        static_assert(WN == 1 || WN == 2, "the waits below name 4 or 8 registers");
                        // the loads are invisible to the compiler's waitcnt pass: tie the results to the wait
                        if constexpr (WN == 2)
                            asm volatile("s_waitcnt vmcnt(0)"
                                         : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]),
                                           "+v"(t[6]), "+v"(t[WN * 4 - 1]) :: "memory");
                        else
                            asm volatile("s_waitcnt vmcnt(0)"
                                         : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]) :: "memory");
#pragma unroll
                        for (int n = 0; n < WN * 4; ++n)
#pragma unroll
                            for (int c = 0; c < 4; ++c) acc[i][n >> 2][4 * (n & 3) + c] += t[n][c];
                    }
                    }
                    __syncthreads();
                    if (tid == 0 && have)
                        __hip_atomic_store(p.sk_flag + vb0 + k, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    e = bound(k + 1);
                    ++k;
                }
                UAVSAL_GEMM_EPILOGUE(1.0f, (smem + (S - 1) * STAGE), false)
            }
#if !UAVSAL_SK_PREFETCH
            if (cur < w1) sg = open_segment(cur);
#endif
        }
        return;
    }
    // Whole-tile path, static walk (tiles slot, slot + gx, ... of the XCD's range).  Tried in round 2 and NOT kept:
    // pulling every tile after a workgroup's first from a per-XCD atomic counter (fetched one tile ahead, so the
    // next tile's first DMA stages are still requested before the epilogue).  2700 tiles over 768 resident
    // workgroups leave CUs with 9..12 tiles, which the dynamic walk evens out -- but end to end it measured 2 %
    // SLOWER (1330 vs 1360 frames/s, same box, two runs each): co-resident workgroups no longer sit on
    // neighbouring tiles, and the kernel carried 21 more spilled SGPRs.
    if (tile >= tile_end) return;
    setup_tile(tile);
    const int npre = nst < D ? nst : D;
    for (int t = 0; t < npre; ++t) issue_tile(t, t);
    while (true) {
        const int m0c = m0, n0c = n0;
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int g = 0; g < 16; ++g) acc[i][j][g] = 0.f;

        int stage = 0, istage = npre % S;
        for (int kt = 0; kt < nst; ++kt) {
            if (kt + D <= nst) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(LPT * (D - 1)) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const bool do_issue = kt + D < nst;
            compute(stage, do_issue, istage);
            if (do_issue) istage = (istage + 1 == S) ? 0 : istage + 1;
            stage = (stage + 1 == S) ? 0 : stage + 1;
        }

        const bool has_next = (tile + tile_step) < tile_end;
        __builtin_amdgcn_s_barrier();            // every wave is done reading the ring (WAR: the
                                                 // next tile's DMA and the epilogue staging reuse it)
        if (has_next) {
            tile += tile_step;
            setup_tile(tile);
            it_kt = 0; it_tap = 0; it_ci = 0;
            for (int t = 0; t < npre; ++t) issue_tile(t, t);
        }

        UAVSAL_GEMM_EPILOGUE(1.0f, (smem + (S - 1) * STAGE), false)
        if (!has_next) break;
    }
}

// =====================================================================================
// Split-fp16 path with PRE-SPLIT operands: LDS-DMA staged, S-stage ring, no conversion work.
//
// The register-staged kernel above re-splits the fp32 activations (hi = fp16(16 x), lo = fp16(16 x - hi)) in
// VALU on every load, once per N tile that needs them, and pays a ds_write for every staged byte.  Here the
// producer of the activation tensor has already written that split (the "split shadow", uavsal_hip.h: per
// pixel and group of 32 channels one 128-byte line [32 hi | 32 lo]), and the weights are packed the same way
// per (K step, output channel), so A and B both go global -> LDS by LDS-DMA exactly as in the fp32 kernel.
// A stage holds two panels [A | B] of rows x 128 B (one K step of 32: hi chunks 0-3, lo chunks 4-7); every
// DMA request fetches 8 rows x one FULL cache line (with planar hi / lo planes each request took half of 16
// lines and L1 <- L2 moved 1.5x the useful bytes -- and that link, ~14 B/clk/CU, is what bounds this kernel:
// profiles/r2_presplit_pmc.md).  Ring kept S-1 stages ahead with counted vmcnt + one raw s_barrier per K
// step; MFMA loop lo*hi + hi*lo + hi*hi on v_mfma_f32_32x32x16_f16.  Every wave issues the same number of DMA
// requests per stage (it owns whole 8-row pieces of both panels), which the counted wait relies on.
// LDS swizzle: 16-byte slot s of row r lives at physical slot s ^ ((r >> 1) & 7): the 16 lanes of every
// ds_read_b128 lane group then touch 16 different (row parity, slot) pairs = all 64 banks once.
template <int WAVES_M, int WAVES_N, int WM, int WN, int TAPS, int S>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64, 2) void conv_gemm_h16_dma_kernel(const ConvK p) {
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int NT = NW * 64;
    constexpr int BM = WAVES_M * WM * 32;
    constexpr int BN = WAVES_N * WN * 32;
    constexpr int KT = 32;                       // channels per K step: one 128-byte line (hi + lo) per row
    constexpr int APAN = BM * 128, BPAN = BN * 128;
    constexpr int STAGE = APAN + BPAN;
    constexpr int AG = BM / 8 / NW, BG = BN / 8 / NW;       // 8-row pieces of A / B a wave stages
    constexpr int PPW = AG + BG;                            // DMA requests per wave and stage
    constexpr int D = S - 1;
    static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0 && NW % 2 == 0, "every wave stages whole 8-row pieces");
    static_assert(PPW * (D > 1 ? D - 1 : 0) < 64, "vmcnt immediate");
    static_assert(STAGE >= 32 * BN * 4, "epilogue staging must fit one ring stage");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

    const int tid = threadIdx.x;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const uavsal_tile_walk walk = xcd_tile_walk(blockIdx.x, gridDim.x, p.nblk);
    int tile = walk.tile;
    const int tile_end = walk.end, tile_step = walk.stride;
    if (tile >= tile_end) return;
    int m0 = 0, n0 = 0;

    // per-lane DMA coordinates: row (lane >> 3) of an 8-row piece, physical 16-byte slot lane & 7.  The swizzle
    // is applied on the SOURCE address: the lane fetches logical slot = physical ^ ((row >> 1) & 7), and since
    // a wave's pieces start at rows 8 * (wave + g * NW) with NW even, (row >> 1) & 7 is the same for all of them.
    const int pr = lane >> 3;
    const int ls = (lane & 7) ^ ((((wave_u & 1) << 2) | (pr >> 1)) & 7);
    long long a_base[AG];            // half offset of the row's centre pixel in the shadow
    int a_taps[AG];
    bool a_ok[AG];
    unsigned b_off[BG];              // byte offset of the row inside one K step's slab of the weights
    bool b_ok[BG];
    const char* zero = reinterpret_cast<const char*>(g_zero16);
    auto setup_tile = [&](int t) {
        const int tile_m = t / p.tiles_n;
        const int tile_n = t - tile_m * p.tiles_n;
        m0 = tile_m * BM;
        n0 = tile_n * BN;
#pragma unroll
        for (int g = 0; g < AG; ++g) {
            const int row = (wave_u + g * NW) * 8 + pr;
            const int m = m0 + row;
            a_ok[g] = m < p.M;
            const int mm = a_ok[g] ? m : 0;
            if (TAPS == 1) {
                a_base[g] = row_off(mm, p.HW, p.a_is, p.contig) * p.ldas;
                a_taps[g] = 0;
            } else {
                const int img = mm / p.HW;
                const int pix = mm - img * p.HW;
                const int y = pix / p.W, x = pix - y * p.W;
                a_base[g] = ((long long)img * p.a_is + pix) * p.ldas;
                int mask = 0;
#pragma unroll
                for (int tp = 0; tp < 9; ++tp) {
                    const int yy = y + tp / 3 - 1, xx = x + tp % 3 - 1;
                    if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) mask |= 1 << tp;
                }
                a_taps[g] = mask;
            }
        }
#pragma unroll
        for (int g = 0; g < BG; ++g) {
            const int n = n0 + (wave_u + g * NW) * 8 + pr;
            b_ok[g] = n < p.Npad;
            b_off[g] = (unsigned)(b_ok[g] ? n : 0) * 128u + ls * 16;
        }
    };

    const int nst = p.ktiles;                    // K steps of 32
    int it_kt = 0, it_tap = 0, it_ci = 0;        // running K position of the next stage to request
    // The PPW requests of a stage are issued as one burst right after the barrier.  That burst blocks the wave
    // for as long as the memory pipe takes to accept the stage (profiles/r2_presplit_pmc.md: DMA-only 345 us +
    // MFMA-only 402 us = 602 us combined at K=1536, i.e. they hardly overlap); slotting the requests between
    // MFMA groups (issue_piece one at a time) was tried and is SLOWER with hipcc's schedule (723 -> 783 us on
    // the 128x128 tile, register spills on 256x256), so the burst stays.
    bool is_kin = false;
    char* is_As = nullptr;
    const char* is_wk = nullptr;
    long long is_aoff = 0;                       // wave-uniform part of the A source offset (halves)
    int is_tap = 0;
    auto begin_stage = [&](int stage) {
        const int kt = it_kt;
        is_kin = kt < nst;
        is_As = smem + stage * STAGE;
        is_tap = it_tap;
        long long tap_off = 0;                   // (dy * W + dx) * ldas
        if (TAPS == 9) {
            const int ty = (is_tap * 11) >> 5;
            tap_off = (long long)((ty - 1) * p.W + (is_tap - ty * 3 - 1)) * p.ldas;
        }
        is_aoff = tap_off + 2 * it_ci;           // channel group ci0 / 32 is 64 halves wide
        is_wk = p.w + (size_t)kt * p.Npad * 128; // [K step][Npad][hi 64 B | lo 64 B]
        ++it_kt;
        if (TAPS == 9) { if (++it_tap == 9) { it_tap = 0; it_ci += KT; } }
        else it_ci += KT;
    };
    auto issue_piece = [&](int q) {              // q: compile-time after unrolling
        if (q < AG) {
            const int g = q;
            bool ok = a_ok[g] && is_kin;
            if (TAPS == 9) ok = ok && ((a_taps[g] >> is_tap) & 1);
            const long long off = a_base[g] + is_aoff + ls * 8;     // slot ls (0-3 hi, 4-7 lo) is 8 halves
            const char* src = ok ? reinterpret_cast<const char*>(p.a_sp + off) : zero;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(is_As + (wave_u + g * NW) * 1024), 16, 0, 0);
        } else {
            const int g = q - AG;
            const char* src = (b_ok[g] && is_kin) ? is_wk + b_off[g] : zero;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(is_As + APAN + (wave_u + g * NW) * 1024), 16, 0, 0);
        }
    };
    auto issue_stage = [&](int stage) {
        begin_stage(stage);
#pragma unroll
        for (int q = 0; q < PPW; ++q) issue_piece(q);
    };

    f32x16 acc[WM][WN];
    const int wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave - wm * WAVES_N;
    const int lr = lane & 31, lh = lane >> 5;

    auto compute_half = [&](int stage, int s) {      // sub-step s (16 of the K step's 32 channels)
        const char* As = smem + stage * STAGE;
        const char* Bs = As + APAN;
        {
            const int chunk = 2 * s + lh;        // hi chunk; the lo chunk is slot chunk + 4
            u32x4 af[WM][2], bfr[WN][2];
#pragma unroll
            for (int i = 0; i < WM; ++i) {
                const int row = (wm * WM + i) * 32 + lr;
                const int sw = (row >> 1) & 7;
                af[i][0] = *reinterpret_cast<const u32x4*>(As + row * 128 + ((chunk ^ sw) << 4));
                af[i][1] = *reinterpret_cast<const u32x4*>(As + row * 128 + (((chunk + 4) ^ sw) << 4));
            }
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const int row = (wn * WN + j) * 32 + lr;
                const int sw = (row >> 1) & 7;
                bfr[j][0] = *reinterpret_cast<const u32x4*>(Bs + row * 128 + ((chunk ^ sw) << 4));
                bfr[j][1] = *reinterpret_cast<const u32x4*>(Bs + row * 128 + (((chunk + 4) ^ sw) << 4));
            }
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    const f16x8 ah = __builtin_bit_cast(f16x8, af[i][0]), al = __builtin_bit_cast(f16x8, af[i][1]);
                    const f16x8 bh = __builtin_bit_cast(f16x8, bfr[j][0]), bl = __builtin_bit_cast(f16x8, bfr[j][1]);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[i][j], 0, 0, 0);
                }
        }
    };

    setup_tile(tile);
    const int npre = nst < D ? nst : D;
    for (int t = 0; t < npre; ++t) issue_stage(t);
    while (true) {
        const int m0c = m0, n0c = n0;
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int g = 0; g < 16; ++g) acc[i][j][g] = 0.f;

        int stage = 0, istage = npre % S;
        for (int kt = 0; kt < nst; ++kt) {
            // this wave's requests for stage kt have landed (the younger D-1 stages may still be in flight)
            if (D > 1 && kt + D <= nst) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PPW * (D > 1 ? D - 1 : 0)) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();        // every wave's have; and stage kt-1 (refilled next) is read out
            // The request burst of a stage blocks the issuing wave for ~2000 cycles (the memory pipe accepts 64 KB
            // no faster), and two waves share each SIMD's matrix pipe.  So the second wave of every SIMD (waves
            // NW/2 .. NW-1) issues its burst in the MIDDLE of the K step: while one wave of a SIMD is blocked the
            // other multiplies (MI355X_MICROARCH.md, two waves per SIMD, item 9).
            const bool do_issue = kt + D < nst;
            const bool late = UAVSAL_H16_STAGGER && NW == 8 && wave_u >= NW / 2;
            if (do_issue && !late) issue_stage(istage);
            compute_half(stage, 0);
            if (do_issue && late) issue_stage(istage);
            compute_half(stage, 1);
            if (do_issue) istage = (istage + 1 == S) ? 0 : istage + 1;
            stage = (stage + 1 == S) ? 0 : stage + 1;
        }

        const bool has_next = (tile + tile_step) < tile_end;
        __builtin_amdgcn_s_barrier();            // ring reads done (WAR for the next tile's DMA / the staging)
        if (has_next) {
            tile += tile_step;
            setup_tile(tile);
            it_kt = 0; it_tap = 0; it_ci = 0;
            for (int t = 0; t < npre; ++t) issue_stage(t);
        }
        UAVSAL_GEMM_EPILOGUE(F16X3_ACC_SCALE, (smem + (S - 1) * STAGE), true)
        if (!has_next) break;
    }
}

// K-split decisions count work units against a FIXED number of workgroup slots (two per CU of a 256-CU part), not against
// what the occupancy query returns on the device at hand: the number of shares, hence the fp32 summation order, is a
// function of the shape only (same results on every SKU / partition mode)
template <int PREC, int WAVES_M, int WAVES_N, int WM, int WN>
int launch_variant(const ConvK& k0, int taps, hipStream_t stream) {
    constexpr int BM = WAVES_M * WM * 32, BN = WAVES_N * WN * 32;
    constexpr int NPAN = (PREC == UAVSAL_PREC_BF16X3 || PREC == UAVSAL_PREC_F16X3) ? 2 : 1;
    constexpr int SMEM = 2 * NPAN * (BM + BN) * 64;
    constexpr int NT = WAVES_M * WAVES_N * 64;
    ConvK k = k0;
    const int tiles_m = (k.M + BM - 1) / BM;
    k.tiles_n = (k.Cout + BN - 1) / BN;
    k.nblk = tiles_m * k.tiles_n;
    if (taps == 1 && k.dw_w) {
        if constexpr (WM * WN <= 4) {      // the fused producer is not built for the 256 x 256 tile
            const int cap = UAVSAL_PER_DEVICE(resident_grid(conv_gemm_kernel<PREC, WAVES_M, WAVES_N, WM, WN, 1, true>, SMEM, NT));
            const int grid = k.nblk < cap ? k.nblk : cap;
            hipLaunchKernelGGL((conv_gemm_kernel<PREC, WAVES_M, WAVES_N, WM, WN, 1, true>), dim3(grid), dim3(NT), SMEM, stream, k);
        } else {
            return UAVSAL_ESHAPE;
        }
    } else if (taps == 1) {
        const int cap = UAVSAL_PER_DEVICE(resident_grid(conv_gemm_kernel<PREC, WAVES_M, WAVES_N, WM, WN, 1>, SMEM, NT));
        // (same K split as the 3x3 tile below: few 64 x 64 tiles, long K -- the 1920 -> 256 ASPP projections at 12x20)
        if (WM * WN == 1 && k.kpart && k.epi == UAVSAL_EPI_AFFINE && (PREC == UAVSAL_PREC_F16X3 || PREC == UAVSAL_PREC_BF16X3)) {
            static const bool on = [] { const char* e = getenv("UAVSAL_SPLITK_1X1"); return !(e && e[0] == '0'); }();
            int ksp = on ? UAVSAL_SPLIT_REF_SLOTS_64 / (k.nblk > 0 ? k.nblk : 1) : 1;
            ksp = ksp >= 4 ? 4 : (ksp >= 2 ? 2 : 1);
            while (ksp > 1 && (k.ktiles % ksp || k.ktiles / ksp < 12)) ksp >>= 1;
            if (ksp > 1 && (long long)ksp * k.M * k.Npad * 4 <= k.kpart_bytes && !(k.Cout & 3) && !(k.ldc & 3)) {
                k.ksplit = ksp;
                k.nblk *= ksp;
            }
        }
        const int grid = k.nblk < cap ? k.nblk : cap;
        hipLaunchKernelGGL((conv_gemm_kernel<PREC, WAVES_M, WAVES_N, WM, WN, 1>), dim3(grid), dim3(NT), SMEM, stream, k);
        if (k.ksplit > 1) return launch_splitk_reduce(k, PREC == UAVSAL_PREC_F16X3 ? F16X3_ACC_SCALE : 1.0f, stream);
    } else {
        const int cap = UAVSAL_PER_DEVICE(resident_grid(conv_gemm_kernel<PREC, WAVES_M, WAVES_N, WM, WN, 9>, SMEM, NT));
        // 64 x 64 tile with fewer tiles than CUs-worth of slots and a long K walk (the ConvTWA step: 228 tiles x 72 K
        // steps): split K over 2 or 4 workgroups per tile; the shares' sums meet in splitk_reduce_kernel
        if (WM * WN == 1 && k.kpart && (k.epi == UAVSAL_EPI_TWA || k.epi == UAVSAL_EPI_AFFINE) &&
            (PREC == UAVSAL_PREC_F16X3 || PREC == UAVSAL_PREC_BF16X3)) {
            static const bool on = [] { const char* e = getenv("UAVSAL_SPLITK_3X3"); return !(e && e[0] == '0'); }();
            int ksp = on ? UAVSAL_SPLIT_REF_SLOTS_64 / (k.nblk > 0 ? k.nblk : 1) : 1;
            ksp = ksp >= 4 ? 4 : (ksp >= 2 ? 2 : 1);
            while (ksp > 1 && (k.ktiles % (9 * ksp) || k.ktiles / ksp < 18)) ksp >>= 1;
            if (ksp > 1 && (long long)ksp * k.M * k.Npad * 4 <= k.kpart_bytes && !(k.Cout & 3) && !(k.ldc & 3) &&
                (k.epi != UAVSAL_EPI_TWA || (!(k.ldx & 3) && !(k.lda & 3) && !(k.ldr & 3)))) {
                k.ksplit = ksp;
                k.nblk *= ksp;
            }
        }
        const int grid = k.nblk < cap ? k.nblk : cap;
        hipLaunchKernelGGL((conv_gemm_kernel<PREC, WAVES_M, WAVES_N, WM, WN, 9>), dim3(grid), dim3(NT), SMEM, stream, k);
        if (k.ksplit > 1) return launch_splitk_reduce(k, PREC == UAVSAL_PREC_F16X3 ? F16X3_ACC_SCALE : 1.0f, stream);
    }
    return uavsal_launch_status();
}

// stream-K is used when whole-tile scheduling would leave part of the chip idle in the last round;
// grid = 2 workgroups per CU (measured best for these shapes), capped at what is resident when the
// launch has the chip to itself.  Progress does not rest on residency (other lanes' kernels take CUs too):
// a workgroup publishes the piece it owes BEFORE it waits, and only waits for higher block ids, so a
// waiting workgroup's producer is either resident (and publishes without waiting) or not yet started
// (and starts as soon as any workgroup -- of any kernel -- retires).  That needs in-order workgroup dispatch,
// which the hardware does and HIP does not promise: hence the bounded wait + error word in the kernel.
static inline int streamk_grid(long long nblk, int kstages, int cus, int cap, bool small_tile) {
    int G = 2 * cus;
    if (G > cap) G = cap;
    if (small_tile) {
        // 64 x 64 tiles (16 KB partials): few tiles with a long serial K loop (the 12x20 / 23x40 maps of
        // the backbone tail) are latency-bound, so K is spread over as many workgroups as get >= 12 stages
        const long long by_len = nblk * kstages / 12;
        if (by_len < G) G = (int)by_len;
        if (G < 2 || nblk < G / 6 + 1 || nblk >= 3LL * cus * 3 || kstages < 24 || nblk % G == 0) return 0;
        const long long per_cu3 = (nblk + 3LL * cus - 1) / (3LL * cus);
        return (double)nblk / (double)(per_cu3 * 3 * cus) < 0.93 ? G : 0;
    }
    // short K loops lose more to publishing / collecting the partial tiles than they win (K=512: 80 -> 90 us,
    // K=768 even; the 2700-tile K=256 expands 227 -> 259), long ones win 8-40 % (profiles/r1_streamk_probe.log)
    if (nblk < G / 4 + 1 || nblk >= 3LL * cus * 3 || kstages < 64 || nblk % G == 0) return 0;
    const long long per_cu3 = (nblk + 3LL * cus - 1) / (3LL * cus);           // rounds at 3 workgroups per CU
    const double eff = (double)nblk / (double)(per_cu3 * 3 * cus);
    return eff < 0.93 ? G : 0;
}

// (block tile, ring) configurations that have a stream-K instance
template <int WAVES_M, int WAVES_N, int WM, int WN, int S, int NKP>
struct SkCfg {
    static constexpr int BM = WAVES_M * WM * 32, BN = WAVES_N * WN * 32, SMEM = S * NKP * (BM + BN) * 64, STAGE_K = 16 * NKP;
    template <int TAPS>
    static int cap() {
        const int c = UAVSAL_PER_DEVICE(resident_grid(conv_gemm_f32_dma_kernel<WAVES_M, WAVES_N, WM, WN, TAPS, S, NKP, true>, SMEM));
        return c;
    }
    template <int TAPS>
    static int launch(const ConvK& k0, int G, hipStream_t stream) {
        ConvK k = k0;
        k.tiles_n = (k.Cout + BN - 1) / BN;
        k.nblk = ((k.M + BM - 1) / BM) * k.tiles_n;
        hipLaunchKernelGGL((conv_gemm_f32_dma_kernel<WAVES_M, WAVES_N, WM, WN, TAPS, S, NKP, true>), dim3(G), dim3(256),
                           SMEM, stream, k);
        return uavsal_launch_status();
    }
};
typedef SkCfg<2, 2, 2, 2, 3, 1> SkBig;      // 128 x 128
typedef SkCfg<2, 2, 1, 1, 3, 2> SkSmall;    // 64 x 64, 32-deep stages: 48 KB of LDS, so three fit a CU
typedef SkCfg<4, 1, 1, 1, 4, 1> SkThin;     // 128 x 32 (Cout <= 32: the 1536 -> 1 decoder projection streams HBM)

template <int WAVES_M, int WAVES_N, int WM, int WN, int S, int NKP>
int launch_f32_dma(const ConvK& k0, int taps, hipStream_t stream) {
    constexpr int BM = WAVES_M * WM * 32, BN = WAVES_N * WN * 32, NT = WAVES_M * WAVES_N * 64;
    constexpr int SMEM = S * NKP * (BM + BN) * 64;
    ConvK k = k0;
    const int tiles_m = (k.M + BM - 1) / BM;
    k.tiles_n = (k.Cout + BN - 1) / BN;
    k.nblk = tiles_m * k.tiles_n;
    if (taps == 1) {
        const int cap = UAVSAL_PER_DEVICE(resident_grid(conv_gemm_f32_dma_kernel<WAVES_M, WAVES_N, WM, WN, 1, S, NKP>, SMEM, NT));
        const int grid = k.nblk < cap ? k.nblk : cap;
        hipLaunchKernelGGL((conv_gemm_f32_dma_kernel<WAVES_M, WAVES_N, WM, WN, 1, S, NKP>), dim3(grid), dim3(NT), SMEM, stream, k);
    } else {
        const int cap = UAVSAL_PER_DEVICE(resident_grid(conv_gemm_f32_dma_kernel<WAVES_M, WAVES_N, WM, WN, 9, S, NKP>, SMEM, NT));
        const int grid = k.nblk < cap ? k.nblk : cap;
        hipLaunchKernelGGL((conv_gemm_f32_dma_kernel<WAVES_M, WAVES_N, WM, WN, 9, S, NKP>), dim3(grid), dim3(NT), SMEM, stream, k);
    }
    return uavsal_launch_status();
}

int launch_f32(const ConvK& k, int taps, int tile, hipStream_t stream) {
    if (tile == 7) return launch_f32_dma<4, 2, 2, 2, 3, 1>(k, taps, stream);     // 256 x 128 on 8 waves, 3 x 24 KB ring
    switch (tile) {
        case 5:
        case 6:
        case 1: return launch_f32_dma<2, 2, 2, 2, 3, 1>(k, taps, stream);    // 128 x 128, 3 x 16 KB ring
        case 2: return launch_f32_dma<4, 1, 1, 2, 4, 1>(k, taps, stream);    // 128 x 64,  4 x 12 KB
        case 3: return launch_f32_dma<4, 1, 1, 1, 4, 1>(k, taps, stream);    // 128 x 32,  4 x 10 KB
        default: return launch_f32_dma<2, 2, 1, 1, 3, 4>(k, taps, stream);   // 64 x 64, 3 x (4 x 8 KB): one
                                                  // workgroup per CU at best, so 64-deep K steps per barrier
    }
}

template <int PREC>
int launch_prec(const ConvK& k, int taps, int tile, hipStream_t stream) {
    if constexpr (PREC == UAVSAL_PREC_BF16X3 || PREC == UAVSAL_PREC_F16X3) {
        // 128 x 256 on 8 waves: one fp32 -> hi/lo conversion of the A tile feeds twice the MFMAs
        if (tile == 5) return launch_variant<PREC, 2, 4, 2, 2>(k, taps, stream);
        // 256 x 256 on 8 waves (128 x 64 per wave): half the L2 -> CU bytes per FLOP of 128 x 256
        if (tile == 6 && !k.dw_w) return launch_variant<PREC, 2, 4, 4, 2>(k, taps, stream);
    }
    switch (tile) {
        case 5:
        case 6:
        case 1: return launch_variant<PREC, 2, 2, 2, 2>(k, taps, stream);   // 128 x 128
        case 2: return launch_variant<PREC, 4, 1, 1, 2>(k, taps, stream);   // 128 x 64
        case 3: return launch_variant<PREC, 4, 1, 1, 1>(k, taps, stream);   // 128 x 32
        default: return launch_variant<PREC, 2, 2, 1, 1>(k, taps, stream);  // 64 x 64
    }
}


template <int WAVES_M, int WAVES_N, int WM, int WN, int S>
int launch_h16_dma(const ConvK& k0, int taps, hipStream_t stream) {
    constexpr int BM = WAVES_M * WM * 32, BN = WAVES_N * WN * 32, NT = WAVES_M * WAVES_N * 64;
    constexpr int SMEM = S * (BM + BN) * 128;
    ConvK k = k0;
    k.tiles_n = (k.Cout + BN - 1) / BN;
    k.nblk = ((k.M + BM - 1) / BM) * k.tiles_n;

    if (taps == 1) {
        const int cap = UAVSAL_PER_DEVICE(resident_grid(conv_gemm_h16_dma_kernel<WAVES_M, WAVES_N, WM, WN, 1, S>, SMEM, NT));
        const int grid = k.nblk < cap ? k.nblk : cap;
        hipLaunchKernelGGL((conv_gemm_h16_dma_kernel<WAVES_M, WAVES_N, WM, WN, 1, S>), dim3(grid), dim3(NT), SMEM, stream, k);
    } else {
        const int cap = UAVSAL_PER_DEVICE(resident_grid(conv_gemm_h16_dma_kernel<WAVES_M, WAVES_N, WM, WN, 9, S>, SMEM, NT));
        const int grid = k.nblk < cap ? k.nblk : cap;
        hipLaunchKernelGGL((conv_gemm_h16_dma_kernel<WAVES_M, WAVES_N, WM, WN, 9, S>), dim3(grid), dim3(NT), SMEM, stream, k);
    }
    return uavsal_launch_status();
}

// LDS-halo depthwise -> projection (fp32 / split-fp16): which descriptors take it, and the launch
bool dwproj_eligible(const uavsal_conv_desc* d) {
    static const bool on = [] { const char* e = getenv("UAVSAL_DWPROJ_LDS"); return !(e && e[0] == '0'); }();
    return on && d->dw_w9c && (d->prec == UAVSAL_PREC_F32 || d->prec == UAVSAL_PREC_F16X3) && d->taps == 1 &&
           d->dw_stride == 1 && d->Cin % 16 == 0 && d->Cin <= UAVSAL_DWPROJ_MAX_C && d->epi == UAVSAL_EPI_AFFINE;
}

// (the kernel and its launcher live in dwproj.hip: uavsal_launch_dwproj)
// pre-split path: 256 x 256 / 128 x 256 on 8 waves, 128 x 128 on 4 (two workgroups per CU)
int launch_h16(const ConvK& k, int taps, int tile, hipStream_t stream) {
    if (tile == 6) return launch_h16_dma<2, 4, 4, 2, 2>(k, taps, stream);    // 2 x 64 KB ring
    if (tile == 5) return launch_h16_dma<2, 4, 2, 2, 3>(k, taps, stream);    // 3 x 48 KB
    return launch_h16_dma<2, 2, 2, 2, 2>(k, taps, stream);                   // 2 x 32 KB
}

// does this descriptor take the pre-split LDS-DMA path (with block tile `tile`)?
bool split_eligible(const uavsal_conv_desc* d, int tile) {
    if (d->prec != UAVSAL_PREC_F16X3 || !d->a_split) return false;
    if (!(tile == 1 || tile == 5 || tile == 6) || (d->Cin % 32) || d->dw_w9c || d->epi != UAVSAL_EPI_AFFINE) return false;
    if ((d->ldas & 63) || d->ldas < 2 * d->Cin || ((uintptr_t)d->a_split & 127)) return false;
    // only the vector epilogue is built for these tiles
    return d->act != UAVSAL_ACT_SIGMOID && !(d->ldc & 3) && !(d->Cout & 3) && uavsal_aligned16(d->out) &&
           (!d->res || (!(d->ldr & 3) && uavsal_aligned16(d->res)));
}

// grid of the stream-K launch this descriptor would get with block tile `tile`, or 0 (whole tiles)
int streamk_plan(const uavsal_conv_desc* d, int tile, int ktiles) {
    if (d->prec != UAVSAL_PREC_F32 || !(tile == 1 || tile == 3 || tile == 4) || d->dw_w9c ||
        d->epi == UAVSAL_EPI_LSTM || !d->sk_ws || !uavsal_aligned16(d->sk_ws))
        return 0;
    const int cus = UAVSAL_PER_DEVICE(([] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }()));
    const long long M = (long long)d->H * d->W * d->n_img;
    const int bm = tile == 4 ? 64 : 128, bn = tile == 1 ? 128 : (tile == 3 ? 32 : 64);
    const long long nblk = ((M + bm - 1) / bm) * ((d->Cout + bn - 1) / bn);
    const bool t1 = d->taps == 1;
    const int cap = tile == 1 ? (t1 ? SkBig::cap<1>() : SkBig::cap<9>())
                  : tile == 3 ? (t1 ? SkThin::cap<1>() : SkThin::cap<9>())
                              : (t1 ? SkSmall::cap<1>() : SkSmall::cap<9>());
    const int kstages = tile == 4 ? (ktiles + 1) / 2 : ktiles;
    const int G = streamk_grid(nblk, kstages, cus, cap, tile != 1);
    if (G <= 0 || G >= UAVSAL_SK_STREAMK_MAX || d->sk_ws_bytes < 65536 + (long long)G * bm * bn * 4) return 0;
    return G;
}

int pick_tile(long long M, int Cout, int prec) {
    // Largest tile that still hands every one of the 256 CUs at least one workgroup: measured on
    // the path's shapes (profiles/r1_gemm_probe.log) 128x128 beats 128x64 / 64x64 as soon as there
    // are >= 256 tiles (K=1536,N=256: 92 vs 73 TFLOP/s; 3x3 448->256: 91 vs 67), because the
    // per-tile L2 traffic per FLOP halves.
    const int npad = (Cout + 31) / 32 * 32;
    if (npad <= 32) return 3;
    if ((prec == UAVSAL_PREC_F16X3 || prec == UAVSAL_PREC_BF16X3) && Cout % 256 == 0) {
        // split 16-bit: the L2 -> CU traffic per FLOP bounds these, so the widest tile that still
        // leaves every CU two workgroups' worth of tiles (profiles/r1_gemm_probe_v3.log)
        if (((M + 255) / 256) * (Cout / 256) >= 512) return 6;
        if (((M + 127) / 128) * (Cout / 256) >= 192) return 5;
    }
    // fp32 256 x 128 on 8 waves (tile 7; a quarter fewer operand bytes per FLOP than 128 x 128) is NOT picked
    // automatically.  Isolated (profiles/r2_f32_tile7_probe.log) the 256 -> 1536 expand gains 3 % at 8 frames and
    // 10 % at 64, the 3x3 448 -> 256 conv 7 %; inside the forward the same build is 0.6 % (1 clip) and 1.1 % (8 clips)
    // SLOWER end to end, two same-box runs each (1444 / 1671 vs 1453 / 1691 frames/s).  UAVSAL_TILE7=1 opts in.
    static const bool tile7_on = [] { const char* e = getenv("UAVSAL_TILE7"); return e && e[0] == '1'; }();
    if (tile7_on && prec == UAVSAL_PREC_F32 && Cout % 128 == 0 && ((M + 255) / 256) * (Cout / 128) >= 768) return 7;
    // Among the tiles that give >= 256 blocks: least padded-N work, the narrower tiles charged 5 / 35 % for
    // their lower efficiency (Cout = 144: 128x64 27.8 us vs 128x128 33.2; 576: 42.2 vs 46.5; 96: 42.2 with
    // 128x128 vs 50.3, same padding; 24: 128x32 19.9 vs 31.1 -- profiles/r1_gemm_probe_v3.log).
    const int cand_bm[4] = {128, 128, 128, 64};
    const int cand_bn[4] = {128, 64, 32, 64};
    const int penalty[4] = {100, 105, 135, 140};
    int best = 0;
    long long best_cost = 0;
    for (int t = 0; t < 4; ++t) {
        const long long tn = (Cout + cand_bn[t] - 1) / cand_bn[t];
        const long long blocks = ((M + cand_bm[t] - 1) / cand_bm[t]) * tn;
        if (blocks < 256) continue;
        const long long cost = tn * cand_bn[t] * penalty[t];
        if (!best || cost < best_cost) { best = t + 1; best_cost = cost; }
    }
    return best ? best : 4;
}

}  // namespace

static int effective_tile(const uavsal_conv_desc* d) {
    if (d->w_group_stride) return d->tile == 11 ? 11 : 8;    // per-image weights: the instances with 32-float K stages, 128 x 128 or 64 x 64
    if (d->n_group) return 11;                               // output-channel groups with their own inputs: the 64 x 64 instance
    int tile = (d->tile >= 1 && d->tile <= 11) ? d->tile
                                             : pick_tile((long long)d->H * d->W * d->n_img, d->Cout, d->prec);
    if (tile == 11 && !uavsal_f32_k32_eligible(d, tile)) tile = 4;
    if (tile == 10 && !uavsal_f32_k32_eligible(d, 10)) tile = 8;
    if ((tile == 8 || tile == 9) && !uavsal_f32_k32_eligible(d, tile)) tile = tile == 9 ? 7 : 1;   // full-line K stages
    // automatic choice: the 128 x 128 launches that do not take the stream-K path move to the kernel with 32-float K
    // stages (conv_gemm_k32.hip); UAVSAL_K32=0 keeps the 16-float one, UAVSAL_K32=9 picks the 256 x 128 instance
    // where it still fills the chip
    // ... and launches with too few tiles for the chip and a long K walk take it with K split over several workgroups
    // per tile (uavsal_f32_k32_ksplit): the ConvTWA step (UAVSAL_K32_SPLITK=1, the default), every such conv (=2), none (=0)
    if (d->tile == 0 && tile == 4 && d->prec == UAVSAL_PREC_F32 && d->sk_ws && d->sk_ws_bytes > 65536) {
        static const int k32_mode = [] { const char* e = getenv("UAVSAL_K32"); return e ? atoi(e) : 1; }();
        static const int sk_mode = [] { const char* e = getenv("UAVSAL_K32_SPLITK"); return e ? atoi(e) : 1; }();
        const bool want = sk_mode == 2 ? (d->epi == UAVSAL_EPI_AFFINE || d->epi == UAVSAL_EPI_TWA) : (sk_mode == 1 && d->epi == UAVSAL_EPI_TWA);
        if (k32_mode && want && uavsal_f32_k32_eligible(d, 8)) {
            const long long M = (long long)d->H * d->W * d->n_img;
            const int npad = (d->Cout + 31) / 32 * 32;
            const int ksp = uavsal_f32_k32_ksplit(((M + 127) / 128) * ((d->Cout + 127) / 128), d->taps * d->Cin / 32);
            // (tile 10 reduces the shares inside the launch -- the last share to arrive adds them -- and measures 12 us
            // SLOWER per ConvTWA step than shares + reduce launch: one workgroup per tile reads all the shares)
            if (ksp > 1 && (long long)ksp * M * npad * 4 <= d->sk_ws_bytes - 65536 && !(d->Cout & 3) && !(d->ldc & 3)) tile = 8;
        }
    }
    // ... and affine convs on the small backbone maps with a long K (the 12x20 projections: at most 160 tiles of 64 x 64,
    // at least 24 stages) take the 64 x 64 instance with K shares over workgroups reduced inside the launch (tile 11):
    // 17.0 / 23.2 / 26.9 / 18.1 us against 19.5 / 26.6 / 32.7 / 25.2 for the stream-K instance (profiles/r3_gemm_k32.md)
    if (d->tile == 0 && tile == 4 && d->prec == UAVSAL_PREC_F32 && d->epi == UAVSAL_EPI_AFFINE && d->sk_ws && d->sk_ws_bytes > 65536) {
        static const int k32_mode = [] { const char* e = getenv("UAVSAL_K32"); return e ? atoi(e) : 1; }();
        static const int small_mode = [] { const char* e = getenv("UAVSAL_K32_SMALL"); return e ? atoi(e) : 1; }();
        const long long M = (long long)d->H * d->W * d->n_img;
        const long long tiles64 = ((M + 63) / 64) * ((d->Cout + 63) / 64);
        // (round 4: from 12 stages on -- features.14's projection, K = 576: 16.9 -> 13.9 us; the context prior's, K = 384 on one
        // 12x20 map: 12.9 -> 10.0)
        if (k32_mode && small_mode && tiles64 <= 160 && d->taps * d->Cin / 32 >= 12 && uavsal_f32_k32_eligible(d, 11) &&
            d->act != UAVSAL_ACT_SIGMOID && !(d->Cout & 3) && !(d->ldc & 3))
            tile = 11;
    }
    // ... and the short-K expands of the small backbone maps (64 -> 384, 96 -> 576, 160 -> 960 on 23x40 / 12x20: at most one
    // round of 64 x 64 tiles at three workgroups per CU) take it too: 9.6 / 15.9 / 11.4 us against 11.1 / 18.4 / 13.0
    if (d->tile == 0 && tile != 11 && d->prec == UAVSAL_PREC_F32 && d->epi == UAVSAL_EPI_AFFINE && d->taps == 1) {
        static const int k32_mode = [] { const char* e = getenv("UAVSAL_K32"); return e ? atoi(e) : 1; }();
        static const int exp_mode = [] { const char* e = getenv("UAVSAL_K32_SMALL_EXPAND"); return e ? atoi(e) : 1; }();
        const long long M = (long long)d->H * d->W * d->n_img;
        const long long tiles64 = ((M + 63) / 64) * ((d->Cout + 63) / 64);
        // (round 4: up to Cin = 256 and two rounds of tiles -- the context prior's expand, 256 -> 1536 on ONE 45x80 map: 40.4 -> 34.2 us)
        if (k32_mode && exp_mode && M <= 8192 && tiles64 <= 1536 && d->Cout >= 256 && d->Cin >= 64 && d->Cin <= 256 &&
            !d->w_group_stride && !d->n_group && uavsal_f32_k32_eligible(d, 11) && d->act != UAVSAL_ACT_SIGMOID && !(d->Cout & 3) && !(d->ldc & 3))
            tile = 11;
    }
    if (d->tile == 0 && tile == 1 && d->prec == UAVSAL_PREC_F32) {
        static const int k32_mode = [] { const char* e = getenv("UAVSAL_K32"); return e ? atoi(e) : 1; }();
        // (K of at least four 32-float stages: at K = 64 the 16-float instance is 1-2 us faster per launch -- three ring
        // stages against two -- and at K = 32 the launch is store-bound either way)
        // (UAVSAL_K32_OVER_SK=1: also where the 16-float instance would run stream-K -- experiment knob)
        static const int over_sk = [] { const char* e = getenv("UAVSAL_K32_OVER_SK"); return e ? atoi(e) : 0; }();
        if (k32_mode && uavsal_f32_k32_eligible(d, 8) && d->epi == UAVSAL_EPI_AFFINE && d->taps * d->Cin >= 128 &&
            (over_sk || streamk_plan(d, 1, (d->taps * d->Cin + 15) / 16) == 0)) {
            const long long M = (long long)d->H * d->W * d->n_img;
            tile = (k32_mode == 9 && d->Cout % 128 == 0 && ((M + 255) / 256) * (d->Cout / 128) >= 1024) ? 9 : 8;
        }
    }
    // ... and a 1x1 whose 128 x 128 tiles are more than one per CU but fewer than the 512 resident slots (the STBlock's
    // 256 -> 256 output conv at one clip: 450 tiles, so most CUs run two and the rest one) takes the 64 x 64 instance: 49.6 -> 46.7 us
    if (d->tile == 0 && tile == 8 && d->prec == UAVSAL_PREC_F32 && d->epi == UAVSAL_EPI_AFFINE && d->taps == 1 && !d->w_group_stride) {
        const long long M = (long long)d->H * d->W * d->n_img;
        const long long tiles128 = ((M + 127) / 128) * ((d->Cout + 127) / 128);
        if (tiles128 > 256 && tiles128 < 512 && d->Cin >= 128 && d->Cout <= 256 && uavsal_f32_k32_eligible(d, 11) &&
            d->act != UAVSAL_ACT_SIGMOID && !(d->Cout & 3) && !(d->ldc & 3))
            tile = 11;
    }
    if (tile == 7 && (d->prec != UAVSAL_PREC_F32 || d->dw_w9c)) tile = 1;     // 256 x 128 exists for the fp32 LDS-DMA kernel
    if (tile == 6) {   // the 256 x 256 tile only carries the vector epilogue
        const bool vec = d->epi == UAVSAL_EPI_AFFINE && d->act != UAVSAL_ACT_SIGMOID && !(d->ldc & 3) &&
                         !(d->Cout & 3) && uavsal_aligned16(d->out) &&
                         (!d->res || (!(d->ldr & 3) && uavsal_aligned16(d->res)));
        if (!vec || d->dw_w9c) tile = 5;
    }
    if (d->epi == UAVSAL_EPI_LSTM) tile = 4;     // the LSTM update lives in the 64x64 tile's vector epilogue
    return tile;
}

extern "C" long long uavsal_streamk_workspace_bytes(void) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    const long long G = 2LL * n;
    return 65536 + G * 128 * 128 * 4;
}

extern "C" int uavsal_conv_streamk_grid(const uavsal_conv_desc* d) {
    if (!d || d->n_img <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0) return 0;
    const int tile = uavsal_conv_tile(d);
    const int KT = d->prec == UAVSAL_PREC_F32 ? 16 : 32;
    return tile > 0 ? streamk_plan(d, tile, (d->taps * d->Cin + KT - 1) / KT) : 0;
}

extern "C" int uavsal_conv_tile(const uavsal_conv_desc* d) {
    if (!d || d->n_img <= 0 || d->H <= 0 || d->W <= 0 || d->Cout <= 0) return UAVSAL_EINVAL;
    return effective_tile(d);
}

extern "C" int uavsal_conv_uses_split(const uavsal_conv_desc* d) {
    if (!d || d->n_img <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0) return 0;
    return split_eligible(d, effective_tile(d)) ? 1 : 0;
}

extern "C" int uavsal_conv_dwproj(const uavsal_conv_desc* d) {
    if (!d || d->n_img <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0 || !dwproj_eligible(d)) return 0;
    return d->Cout > 128 ? 256 : (d->Cout > 64 ? 128 : (d->Cout > 32 ? 64 : 32));
}

extern "C" int uavsal_conv_gemm(const uavsal_conv_desc* d, uavsal_stream_t stream) {
    if (!d || !d->w || !d->out) return UAVSAL_EINVAL;
    if (!d->a && !(d->a_split && d->epi == UAVSAL_EPI_AFFINE)) return UAVSAL_EINVAL;
    if (d->n_img <= 0 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0) return UAVSAL_EINVAL;
    if (d->taps != 1 && d->taps != 9) return UAVSAL_ESHAPE;
    if (d->prec < 0 || d->prec > 3) return UAVSAL_ESHAPE;
    if (d->Cin & 3) return UAVSAL_EALIGN;
    if (d->a && ((d->lda & 3) || d->lda < d->Cin || !uavsal_aligned16(d->a))) return UAVSAL_EALIGN;
    if (d->epi != UAVSAL_EPI_LSTM && d->ldc < d->Cout) return UAVSAL_ESHAPE;
    if (!uavsal_aligned16(d->w)) return UAVSAL_EALIGN;
    if (d->out_split && ((d->ldos & 63) || d->ldos < 2 * d->Cout || ((uintptr_t)d->out_split & 127)))
        return UAVSAL_EALIGN;
    if (d->out_split && d->prec != UAVSAL_PREC_F16X3) return UAVSAL_ESHAPE;   // only those kernels carry the store
    if (d->out_split) {  // the shadow is written by the vector epilogue only
        const int t = effective_tile(d);
        const bool twa_vec = d->epi == UAVSAL_EPI_TWA && (t == 3 || t == 4) && !(d->ldx & 3) && !(d->lda & 3) &&
                             uavsal_aligned16(d->aux);
        const bool vec = (d->epi == UAVSAL_EPI_AFFINE || twa_vec) && d->act != UAVSAL_ACT_SIGMOID && !(d->ldc & 3) &&
                         !(d->Cout & 3) && uavsal_aligned16(d->out) && (!d->res || (!(d->ldr & 3) && uavsal_aligned16(d->res)));
        if (!vec) return UAVSAL_ESHAPE;
    }
    if (d->taps == 9 && (d->Cin % 32)) return UAVSAL_ESHAPE;
    if ((d->scale == nullptr) != (d->bias == nullptr)) return UAVSAL_EINVAL;
    if (d->res && d->epi != UAVSAL_EPI_LSTM && d->ldr < d->Cout) return UAVSAL_ESHAPE;
    if (d->epi == UAVSAL_EPI_TWA) {
        if (!d->res || !d->aux || d->Cin != d->Cout || d->ldx < d->Cout) return UAVSAL_ESHAPE;
    } else if (d->epi == UAVSAL_EPI_LSTM) {
        if (!d->res || !d->aux || !d->out2 || (d->Cout & 3) || d->Cin * 4 != d->Cout) return UAVSAL_ESHAPE;
        if (d->ldc < d->Cin || d->ld2 < d->Cin || d->ldr < d->Cin || d->ldx < d->Cout) return UAVSAL_ESHAPE;
        if ((d->ldx & 3) || (d->ldc & 3) || (d->ldr & 3) || !uavsal_aligned16(d->aux) ||
            !uavsal_aligned16(d->out) || !uavsal_aligned16(d->res)) return UAVSAL_EALIGN;
    } else if (d->epi != UAVSAL_EPI_AFFINE) {
        return UAVSAL_ESHAPE;
    }
    const long long HW = (long long)d->H * d->W;
    const long long M = HW * d->n_img;
    if (M > 0x7fffffffLL) return UAVSAL_ESHAPE;
    if (d->dw_w9c) {     // fused depthwise producer
        if (d->taps != 1 || d->epi != UAVSAL_EPI_AFFINE || !d->dw_scale || !d->dw_bias) return UAVSAL_ESHAPE;
        if (d->dw_stride != 1 && d->dw_stride != 2) return UAVSAL_ESHAPE;
        if ((d->dw_Hin - 1) / d->dw_stride + 1 != d->H || (d->dw_Win - 1) / d->dw_stride + 1 != d->W) return UAVSAL_ESHAPE;
        if (d->a_img_stride < (long long)d->dw_Hin * d->dw_Win) return UAVSAL_ESHAPE;
        if (!uavsal_aligned16(d->dw_w9c) || !uavsal_aligned16(d->dw_scale) || !uavsal_aligned16(d->dw_bias)) return UAVSAL_EALIGN;
    } else if (d->a_img_stride < HW) {
        return UAVSAL_ESHAPE;
    }
    if (d->o_img_stride < HW) return UAVSAL_ESHAPE;

    ConvK k;
    k.a = d->a; k.w = (const char*)d->w; k.scale = d->scale; k.bias = d->bias;
    k.out = d->out; k.res = d->res; k.aux = d->aux; k.out2 = d->out2; k.ld2 = d->ld2;
    k.dw_w = d->dw_w9c; k.dw_s = d->dw_scale; k.dw_b = d->dw_bias;
    k.dw_stride = d->dw_stride; k.Hin = d->dw_Hin; k.Win = d->dw_Win;
    k.a_is = d->a_img_stride; k.o_is = d->o_img_stride;
    k.r_is = d->res ? d->r_img_stride : HW; k.x_is = d->aux ? d->x_img_stride : HW;
    k.lda = d->lda; k.ldc = d->ldc; k.ldr = d->ldr; k.ldx = d->ldx;
    k.M = (int)M; k.HW = (int)HW; k.H = d->H; k.W = d->W; k.Cin = d->Cin; k.Cout = d->Cout;
    const int KT = d->prec == UAVSAL_PREC_F32 ? 16 : 32;
    k.Kpad = (d->taps * d->Cin + KT - 1) / KT * KT;
    k.Npad = (d->Cout + 31) / 32 * 32;
    k.ktiles = k.Kpad / KT;
    if ((long long)k.Npad * k.Kpad * 4 > 0x7fffffffLL) return UAVSAL_ESHAPE;   // 32-bit weight row offsets
    k.act = d->act; k.epi = d->epi;
    k.contig = ((k.a_is == HW || k.dw_w) && k.o_is == HW && k.r_is == HW && k.x_is == HW) ? 1 : 0;
    k.tiles_n = 0; k.nblk = 0;
    k.sk_part = nullptr; k.sk_flag = nullptr; k.err = nullptr; k.ksplit = 1; k.kpart = nullptr; k.kpart_bytes = 0;
    k.sk_spin = d->sk_spin_limit > 0 ? d->sk_spin_limit : (1 << 22);
    k.sk_drop = d->sk_debug_drop;
    const int tile = effective_tile(d);
    hipStream_t s = (hipStream_t)stream;
    k.w_gs = d->w_group_stride;
    if (k.w_gs && !((tile == 8 || tile == 11) && uavsal_f32_k32_eligible(d, tile) && d->epi == UAVSAL_EPI_AFFINE && !d->a_split))
        return UAVSAL_ESHAPE;
    k.ngrp = d->n_group; k.a_goff = d->a_group_off;
    if (d->n_group < 0 || (k.ngrp && !(uavsal_f32_k32_eligible(d, 11) && !d->a_split && !d->out_split))) return UAVSAL_ESHAPE;
    k.a_sp = (const _Float16*)d->a_split; k.ldas = d->ldas;
    k.out_sp = (_Float16*)d->out_split; k.ldos = d->ldos;
    if (split_eligible(d, tile)) return launch_h16(k, d->taps, tile, s);
    // K-split workspace: the partial-tile area of the caller's stream-K workspace (launches on one lane are ordered)
    if (d->sk_ws && uavsal_aligned16(d->sk_ws) && d->sk_ws_bytes > 65536) {
        k.kpart = (float*)((char*)d->sk_ws + 65536);
        k.kpart_bytes = d->sk_ws_bytes - 65536;
    }
    if (dwproj_eligible(d))
        return uavsal_launch_dwproj(k, d->prec, s);
    if (!d->a) return UAVSAL_EINVAL;             // pre-split operands given but the shape is not eligible
    {
        const int G = streamk_plan(d, tile, k.ktiles);
        if (G > 0) {
            // workspace: 64 KB of flags (one per workgroup + a "wait gave up" word), then the partial tiles
            k.sk_flag = (int*)d->sk_ws;
            k.sk_part = (float*)((char*)d->sk_ws + 65536);
            k.err = d->err ? d->err : (int*)d->sk_ws + UAVSAL_SK_ERR_WORD;
            const bool t1 = d->taps == 1;
            if (tile == 1) return t1 ? SkBig::launch<1>(k, G, s) : SkBig::launch<9>(k, G, s);
            if (tile == 3) return t1 ? SkThin::launch<1>(k, G, s) : SkThin::launch<9>(k, G, s);
            return t1 ? SkSmall::launch<1>(k, G, s) : SkSmall::launch<9>(k, G, s);
        }
    }
    if (tile >= 8 && tile <= 11) {
        if (k.kpart) k.sk_flag = (int*)d->sk_ws + UAVSAL_SK_TICKET_BASE;       // per-tile ticket counters of the K-split launch: their own region
        return uavsal_launch_f32_k32(k, d->taps, tile, s);
    }
    switch (d->prec) {
        case UAVSAL_PREC_F32:    // the fused producer needs register staging: use the generic kernel
            return k.dw_w ? launch_prec<UAVSAL_PREC_F32>(k, d->taps, tile, s) : launch_f32(k, d->taps, tile, s);
        case UAVSAL_PREC_BF16X3: return launch_prec<UAVSAL_PREC_BF16X3>(k, d->taps, tile, s);
        case UAVSAL_PREC_F16X3: return launch_prec<UAVSAL_PREC_F16X3>(k, d->taps, tile, s);
        default: return launch_prec<UAVSAL_PREC_BF16>(k, d->taps, tile, s);
    }
}

// fp32 implicit GEMM (dense 1x1 / 3x3 conv, exact fp32 MFMA) with FULL-LINE K stages.
//
// Same contract, operand layouts, tile walk and epilogue as conv_gemm_f32_dma_kernel (conv_gemm.hip; reference
// model.py:65-72, 89, 94); what differs is how the operands travel:
//   * a K stage is 32 floats = one 128-byte cache line per tile row.  The 16-float stages of the older kernel ask
//     L2 for half a line per row and stage; the L1 fills whole lines (TCP_TCC_READ_REQ counts 128-byte requests),
//     the other half is evicted before the next stage wants it (three workgroups' panels = 48 KB of lines against
//     a 32 KB L1), so the L2 -> L1 link carried twice the operand bytes: 16 B/clk/CU at the full matrix rate
//     against the ~13-14 B/clk/CU it delivers.  One DMA request now fetches 8 rows x one whole line.
//   * panel rows are 128 B in LDS; 16-byte slot s of row r sits at physical slot s ^ ((r >> 1) & 7): the 16 lanes
//     of every ds_read_b128 lane group touch all 64 banks once (the swizzle is applied on the per-lane SOURCE
//     address, the LDS image of a request is lane-linear).
//   * two-stage ring, 64 KB (128 x 128 tile) -> two workgroups per CU; the requests of stage kt + 1 are issued one
//     behind each of the first MFMA groups of stage kt, with running per-lane source pointers (1x1: one 64-bit
//     add per request), so every request has at least half a stage of matrix work to land behind.
//   * barriers are `s_barrier` in inline asm: hipcc turns __builtin_amdgcn_s_barrier() into a full
//     `s_waitcnt vmcnt(0)` drain on gfx950 (no back-off barrier), which is what kept the older kernel's three-stage
//     ring from ever being more than one stage ahead.
//   * fragments of K sub-step u + 1 are read while sub-step u multiplies.
// Eligible: fp32, Cin % 32 == 0, Cin <= 4096, no fused depthwise producer (host: uavsal_f32_k32_eligible).
#define UAVSAL_EPI_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#include "conv_gemm_common.h"

namespace {

template <int WAVES_M, int WAVES_N, int TAPS, int MINW>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64, MINW) void conv_gemm_f32_k32_kernel(const ConvK p) {
    constexpr int WM = 2, WN = 2;
    constexpr int BM = WAVES_M * WM * 32, BN = WAVES_N * WN * 32;
    constexpr int KT = 32;
    constexpr int NW = WAVES_M * WAVES_N, NT = NW * 64;
    constexpr int RPI = NT / 8;                  // panel rows one request round of the workgroup covers
    constexpr int A_IT = BM / RPI, B_IT = BN / RPI, LPT = A_IT + B_IT;
    constexpr int APAN = BM * 128, BPAN = BN * 128, STAGE = APAN + BPAN;
    constexpr int NGRP = 4 * WM * WN;            // MFMA groups (4 MFMAs each) of one stage and wave
    static_assert(BM % RPI == 0 && BN % RPI == 0, "panels divide over the workgroup");
    static_assert(RPI % 16 == 0, "the swizzle term of a lane is the same in every request round");
    static_assert(LPT <= NGRP, "one request behind each MFMA group");
    static_assert(STAGE >= 32 * BN * 4, "epilogue staging must fit one ring stage");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

    const int tid = threadIdx.x;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int wm = wave_u / WAVES_N, wn = wave_u - wm * WAVES_N;
    const int lr = lane & 31, lh = lane >> 5;
    const uavsal_tile_walk walk = xcd_tile_walk(blockIdx.x, gridDim.x, p.nblk);
    int tile = walk.tile;
    if (tile >= walk.end) return;
    int m0 = 0, n0 = 0;

    // request coordinates of this lane: row (tid >> 3) + it * RPI of a panel, physical slot tid & 7
    const int r8 = tid >> 3;
    const int lc = (tid & 7) ^ ((r8 >> 1) & 7);          // logical 16-byte chunk of the row this lane fetches
    const float* a_ptr[A_IT];                            // 1x1: running source pointers (32 floats further per stage)
    const float* b_ptr[B_IT];
    long long a_base[A_IT];                              // 3x3: element offset of the row's centre pixel ...
    int a_taps[A_IT];                                    // ... and which of its nine taps lie inside the image
    const int nst = p.Kpad / KT;
    int it_tap = 0, it_cb = 0;                           // 3x3: tap / channel block of the next stage to request

    auto setup_tile = [&](int t) {
        const int tile_m = t / p.tiles_n;
        const int tile_n = t - tile_m * p.tiles_n;
        m0 = tile_m * BM;
        n0 = tile_n * BN;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int m = m0 + r8 + it * RPI;
            const bool ok = m < p.M;
            if (TAPS == 1) {
                // rows past M walk a row of zeros (their accumulators are never stored)
                a_ptr[it] = ok ? p.a + row_off(m, p.HW, p.a_is, p.contig) * p.lda + lc * 4 : g_zero_row + lc * 4;
            } else {
                const int mm = ok ? m : 0;
                const int img = mm / p.HW;
                const int pix = mm - img * p.HW;
                const int y = pix / p.W, x = pix - y * p.W;
                a_base[it] = ((long long)img * p.a_is + pix) * p.lda + lc * 4;
                int mask = 0;
#pragma unroll
                for (int tp = 0; tp < 9; ++tp) {
                    const int yy = y + tp / 3 - 1, xx = x + tp % 3 - 1;
                    if (ok && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) mask |= 1 << tp;
                }
                a_taps[it] = mask;
            }
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            // weight rows past Npad re-read the last real row (those columns are never stored)
            const int nn = min(n0 + r8 + it * RPI, p.Npad - 1);
            b_ptr[it] = reinterpret_cast<const float*>(p.w) + (size_t)nn * p.Kpad + lc * 4;
        }
        it_tap = 0; it_cb = 0;
    };

    // request r (compile-time) of the next stage, into the ring stage at `st`
    long long tap_off = 0;
    int tap_bit = 0;
    auto next_stage = [&]() {                            // 3x3: advance to the stage about to be requested
        if (TAPS == 9) {
            const int ty = (it_tap * 11) >> 5;           // tap / 3
            tap_off = (long long)((ty - 1) * p.W + (it_tap - ty * 3 - 1)) * p.lda + it_cb * KT;
            tap_bit = it_tap;
            if (++it_tap == 9) { it_tap = 0; ++it_cb; }
        }
    };
    auto issue_one = [&](int r, char* st) {
        if (r < A_IT) {
            const float* src;
            if (TAPS == 1) {
                src = a_ptr[r];
                a_ptr[r] += KT;
            } else {
                src = ((a_taps[r] >> tap_bit) & 1) ? p.a + a_base[r] + tap_off : g_zero16;
            }
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(st + (r * RPI + wave_u * 8) * 128), 16, 0, 0);
        } else {
            const int it = r - A_IT;
            __builtin_amdgcn_global_load_lds((gptr_t)b_ptr[it], (lptr_t)(st + APAN + (it * RPI + wave_u * 8) * 128), 16, 0, 0);
            b_ptr[it] += KT;
        }
    };
    auto issue_stage = [&](char* st) {
        next_stage();
#pragma unroll
        for (int r = 0; r < LPT; ++r) issue_one(r, st);
    };

    // fragment addressing: row (wm * WM + i) * 32 + lr of the A panel, chunk 2u + lh at slot chunk ^ ((lr >> 1) & 7)
    const int sw = (lr >> 1) & 7;
    const int a_row = (wm * WM * 32 + lr) * 128, b_row = APAN + (wn * WN * 32 + lr) * 128;
    f32x16 acc[WM][WN];
    f32x4 fa[2][WM], fb[2][WN];
    auto ldfrag = [&](const char* st, int u, int buf) {
        const int so = ((2 * u + lh) ^ sw) * 16;
#pragma unroll
        for (int i = 0; i < WM; ++i) fa[buf][i] = *reinterpret_cast<const f32x4*>(st + a_row + i * 32 * 128 + so);
#pragma unroll
        for (int j = 0; j < WN; ++j) fb[buf][j] = *reinterpret_cast<const f32x4*>(st + b_row + j * 32 * 128 + so);
    };
    // one stage: 4 K sub-steps x WM x WN groups of 4 MFMAs; ISSUE: the next stage's requests go out one behind each
    // of the first LPT groups
    auto stage = [&](auto issue, const char* st, char* ist) {
        constexpr bool ISSUE = decltype(issue)::value;
        if (ISSUE) next_stage();
        ldfrag(st, 0, 0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (u + 1 < 4) ldfrag(st, u + 1, (u + 1) & 1);
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    const f32x4 av = fa[u & 1][i], bv = fb[u & 1][j];
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[i][j], 0, 0, 0);
                    const int g = (u * WM + i) * WN + j;
                    if (ISSUE && g < LPT) issue_one(g, ist);
                }
        }
    };

    setup_tile(tile);
    issue_stage(smem);
    while (true) {
        const int m0c = m0, n0c = n0;
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int g = 0; g < 16; ++g) acc[i][j][g] = 0.f;

        // stage kt lives in ring slot kt & 1.  Per stage: own requests landed, own LDS reads retired, then everyone's
        // (the barrier also frees the other slot: every wave is past its reads of stage kt - 1).
        int kt = 0;
        for (; kt + 1 < nst; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            char* st = smem + (kt & 1) * STAGE;
            stage(std::true_type{}, st, smem + ((kt & 1) ^ 1) * STAGE);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        stage(std::false_type{}, smem + (kt & 1) * STAGE, nullptr);

        const bool has_next = (tile + walk.stride) < walk.end;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // every wave is past its reads of the ring
        if (has_next) {
            tile += walk.stride;
            setup_tile(tile);
            issue_stage(smem);                   // next tile's stage 0 -> slot 0; the epilogue stages through slot 1
        }
        UAVSAL_GEMM_EPILOGUE(1.0f, (smem + STAGE), false)
        if (!has_next) break;
    }
}

template <int WAVES_M, int WAVES_N, int MINW>
int launch_k32(const ConvK& k0, int taps, hipStream_t stream) {
    constexpr int BM = WAVES_M * 64, BN = WAVES_N * 64, NT = WAVES_M * WAVES_N * 64;
    constexpr int SMEM = 2 * (BM + BN) * 128;
    ConvK k = k0;
    k.tiles_n = (k.Cout + BN - 1) / BN;
    k.nblk = ((k.M + BM - 1) / BM) * k.tiles_n;
    auto cap_of = [](auto kernel) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        int per_cu = 0, cus = 0, dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 256;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, NT, SMEM) != hipSuccess || per_cu <= 0) per_cu = 1;
        return per_cu * cus;
    };
    if (taps == 1) {
        static const int cap = cap_of(conv_gemm_f32_k32_kernel<WAVES_M, WAVES_N, 1, MINW>);
        const int grid = k.nblk < cap ? k.nblk : cap;
        hipLaunchKernelGGL((conv_gemm_f32_k32_kernel<WAVES_M, WAVES_N, 1, MINW>), dim3(grid), dim3(NT), SMEM, stream, k);
    } else {
        static const int cap = cap_of(conv_gemm_f32_k32_kernel<WAVES_M, WAVES_N, 9, MINW>);
        const int grid = k.nblk < cap ? k.nblk : cap;
        hipLaunchKernelGGL((conv_gemm_f32_k32_kernel<WAVES_M, WAVES_N, 9, MINW>), dim3(grid), dim3(NT), SMEM, stream, k);
    }
    return uavsal_launch_status();
}

}  // namespace

__attribute__((visibility("hidden"))) bool uavsal_f32_k32_eligible(const uavsal_conv_desc* d, int tile) {
    if (tile != 8 && tile != 9) return false;
    if (d->prec != UAVSAL_PREC_F32 || d->dw_w9c || d->epi == UAVSAL_EPI_LSTM) return false;
    if ((d->Cin % 32) || d->Cin > UAVSAL_DWPROJ_MAX_C) return false;
    return d->taps == 1 || d->taps == 9;
}

__attribute__((visibility("hidden"))) int uavsal_launch_f32_k32(const uavsal_gemm::ConvK& k, int taps, int tile,
                                                                 hipStream_t stream) {
    if (tile == 9) return launch_k32<4, 2, 2>(k, taps, stream);     // 256 x 128 on 8 waves, one workgroup per CU
    return launch_k32<2, 2, 2>(k, taps, stream);                    // 128 x 128 on 4 waves, two per CU
}

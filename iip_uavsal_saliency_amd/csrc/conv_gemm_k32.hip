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

// Diagnostic builds only (tools/build_probe.sh -> libuavsal_hip_probe.so; the product build has none of this):
//   -DUAVSAL_PROBE: act = 100 + bits {1 no output store, 2 no MFMAs, 4 no DMA requests, 8 no fragment reads,
//                   16 no epilogue} times the kernel with parts compiled out (tools/k32_probe.py parts);
//   -DUAVSAL_K32_STAMPS: every wave sums the cycles (s_memtime) it spends waiting at the stage head, in the stage
//                   body, between the K loop and the epilogue, and in the epilogue, into p.kpart[0..4] (u64).
#ifdef UAVSAL_PROBE
#undef UAVSAL_STORE_OK
#define UAVSAL_STORE_OK(act) (((act) < 100) || !(((act) - 100) & 1))
#endif
#ifdef UAVSAL_K32_STAMPS
#define K32_STAMP(var) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); var += t_ - t_prev; t_prev = t_; }
#else
#define K32_STAMP(var)
#endif

// Experiment knobs (tools/build_probe.sh builds one library per setting; defaults = the product build)
#ifndef K32_PRIO
#define K32_PRIO 0          // 1: the workgroup in the even wave slot of a SIMD runs at raised priority (its CU-mate fills the gaps)
#endif
#ifndef K32_STAGGER
#define K32_STAGGER 0       // N > 0: the workgroup in the odd wave slot starts N x 64 cycles late (s_sleep)
#endif
#ifndef K32_FRAG
#define K32_FRAG 0          // 1: pin the fragment reads of sub-step u + 1 behind the first MFMAs of sub-step u
#endif

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
#ifdef UAVSAL_PROBE
    const int pr_bits = p.act >= 100 ? p.act - 100 : 0;
    const bool pr_mul = !(pr_bits & 2), pr_dma = !(pr_bits & 4), pr_frag = !(pr_bits & 8), pr_epi = !(pr_bits & 16);
#else
    constexpr bool pr_mul = true, pr_dma = true, pr_frag = true, pr_epi = true;
#endif
#ifdef UAVSAL_K32_STAMPS
    unsigned long long t_prev = __builtin_amdgcn_s_memtime(), c_wait = 0, c_body = 0, c_tail = 0, c_epi = 0;
    const unsigned long long t_begin = t_prev;
#endif
    const uavsal_tile_walk walk = xcd_tile_walk(blockIdx.x, gridDim.x, p.nblk);
    int tile = walk.tile;
    if (tile >= walk.end) return;
    int m0 = 0, n0 = 0;
#if K32_PRIO || K32_STAGGER
    // HW_REG_HW_ID[3:0] = wave slot on the SIMD: the two co-resident workgroups of a CU sit in different slots
    const int slot_odd = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 1;
#if K32_PRIO
    if (!slot_odd) __builtin_amdgcn_s_setprio(2);
#endif
#if K32_STAGGER
    if (slot_odd) __builtin_amdgcn_s_sleep(K32_STAGGER);
#endif
#endif

    // request coordinates of this lane: row (tid >> 3) + it * RPI of a panel, physical slot tid & 7
    const int r8 = tid >> 3;
    const int lc = (tid & 7) ^ ((r8 >> 1) & 7);          // logical 16-byte chunk of the row this lane fetches
    const float* a_ptr[A_IT];                            // 1x1: running source pointers (32 floats further per stage)
    const float* b_ptr[B_IT];
    long long a_base[A_IT];                              // 3x3: element offset of the row's centre pixel ...
    int a_taps[A_IT];                                    // ... and which of its nine taps lie inside the image
    const int nst_all = p.Kpad / KT;
    int nst = nst_all;                                   // stages of the current tile (a share of them when K is split)
    int ks = 0;                                          // ... and which share
    int it_tap = 0, it_cb = 0;                           // 3x3: tap / channel block of the next stage to request

    auto setup_tile = [&](int t) {
        int s0 = 0;                                      // K split: t = tile * ksplit + share
        if (p.ksplit > 1) {
            ks = t % p.ksplit;
            t /= p.ksplit;
            s0 = ks * nst_all / p.ksplit;
            nst = (ks + 1) * nst_all / p.ksplit - s0;
        }
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
                a_ptr[it] = (ok ? p.a + row_off(m, p.HW, p.a_is, p.contig) * p.lda + lc * 4 : g_zero_row + lc * 4) + s0 * KT;
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
            // (per-image weights: the tile lies inside image m0 / HW -- HW is a multiple of the tile height)
            b_ptr[it] = reinterpret_cast<const float*>(p.w) + (p.w_gs ? (size_t)(m0 / p.HW) * p.w_gs : 0) +
                        (size_t)nn * p.Kpad + lc * 4 + s0 * KT;
        }
        it_cb = s0 / 9; it_tap = s0 - it_cb * 9;         // (3x3: stage s = channel block s / 9, tap s % 9)
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
        if (!pr_dma) return;
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
        if (!pr_frag) return;
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
                    if (pr_mul) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[i][j], 0, 0, 0);
                    }
                    const int g = (u * WM + i) * WN + j;
                    if (ISSUE && g < LPT) issue_one(g, ist);
                }
#if K32_FRAG
            // order within the sub-step: 2 MFMAs, the 4 fragment reads of the next sub-step, the other 14 MFMAs
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, WM + WN, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * WM * WN - 2, 0);
#endif
        }
    };

    setup_tile(tile);
    issue_stage(smem);
    while (true) {
        const int m0c = m0, n0c = n0, ksc = ks;
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
            K32_STAMP(c_wait)
            char* st = smem + (kt & 1) * STAGE;
            stage(std::true_type{}, st, smem + ((kt & 1) ^ 1) * STAGE);
            K32_STAMP(c_body)
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        K32_STAMP(c_wait)
        stage(std::false_type{}, smem + (kt & 1) * STAGE, nullptr);
        K32_STAMP(c_body)

        const bool has_next = (tile + walk.stride) < walk.end;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // every wave is past its reads of the ring
        if (has_next) {
            tile += walk.stride;
            setup_tile(tile);
            issue_stage(smem);                   // next tile's stage 0 -> slot 0; the epilogue stages through slot 1
        }
        K32_STAMP(c_tail)
        if (p.ksplit > 1) {
            // a K share: raw sums -> p.kpart[share][M][Npad] (splitk_reduce_kernel adds the shares in a fixed order and
            // applies the epilogue).  Barrier-free: every wave passes its 32 x 32 blocks through its own 4 KB of slot 1
            float* stg = reinterpret_cast<float*>(smem + STAGE) + wave_u * 1024;
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) stg[((q & 3) + 8 * (q >> 2) + 4 * lh) * 32 + lr] = acc[i][j][q];
                    const int gn = n0c + (wn * WN + j) * 32 + (lane & 7) * 4;
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int row = (lane >> 3) + it * 8;
                        const int gm = m0c + (wm * WM + i) * 32 + row;
                        const f32x4 v = *reinterpret_cast<const f32x4*>(stg + row * 32 + (lane & 7) * 4);
                        if (gm < p.M && gn < p.Npad)
                            *reinterpret_cast<f32x4*>(p.kpart + ((size_t)ksc * p.M + gm) * p.Npad + gn) = v;
                    }
                }
        } else if (pr_epi) UAVSAL_GEMM_EPILOGUE(1.0f, (smem + STAGE), false)
        K32_STAMP(c_epi)
        if (!has_next) break;
    }
#ifdef UAVSAL_K32_STAMPS
    if (lane == 0 && p.kpart) {
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(p.kpart);
        atomicAdd(dbg + 0, c_wait); atomicAdd(dbg + 1, c_body); atomicAdd(dbg + 2, c_tail); atomicAdd(dbg + 3, c_epi);
        atomicAdd(dbg + 4, __builtin_amdgcn_s_memtime() - t_begin); atomicAdd(dbg + 5, 1ull);
    }
#endif
}


// ---------------------------------------------------------------------------------------------------------
// The same GEMM as ONE continuous stream of K stages over all the tiles of a workgroup ("flat pipeline").
//   * one barrier per stage, placed where nothing is pending: in the middle of the stage's LAST K sub-step.  By then
//     the requests of stage g + 1 are a whole stage old (`vmcnt(0)` costs nothing) and every fragment of stage g is in
//     registers, so the barrier at once publishes stage g + 1 (RAW) and frees the slot of stage g (WAR);
//   * behind it the wave reads the first fragments of stage g + 1 -- they land under the remaining MFMAs of stage g,
//     there is no read bubble at a stage head -- and requests stage g + 2 into the freed slot: a two-slot ring that is
//     a full stage ahead;
//   * the stream runs across tile boundaries: while a tile's epilogue stores, the next tile's first two stages are in
//     flight and its first fragments are already in registers (the epilogue stages through its own 16 KB);
//   * fragment reads are `ds_read_b128` in inline asm with counted `lgkmcnt` waits (hipcc re-merges the two fragment
//     sets of a compiler-scheduled loop into one and waits `lgkmcnt(0)` in front of every sub-step).
// SPLIT: the launch's work units are (tile, K share) pairs, p.ksplit shares per tile (launches with few tiles and a
// long K: the ConvTWA step).  A share publishes its raw sums (write-through stores), takes a ticket on the tile's
// counter, and the share that arrives LAST adds all of them in share order -- the order is fixed, whoever reduces --
// and applies the epilogue (BN / activation / residual or the ConvTWA update, model_convlstm.py:276-292).  Nobody waits.
template <int WAVES_M, int WAVES_N, int TAPS, int MINW, bool SPLIT = false>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64, MINW) void conv_gemm_f32_k32p_kernel(const ConvK p) {
    constexpr int WM = 2, WN = 2;
    constexpr int BM = WAVES_M * WM * 32, BN = WAVES_N * WN * 32;
    constexpr int KT = 32;
    constexpr int NW = WAVES_M * WAVES_N, NT = NW * 64;
    constexpr int RPI = NT / 8;
    constexpr int A_IT = BM / RPI, B_IT = BN / RPI, LPT = A_IT + B_IT;
    constexpr int APAN = BM * 128, BPAN = BN * 128, STAGE = APAN + BPAN;
    static_assert(BM % RPI == 0 && BN % RPI == 0 && RPI % 16 == 0, "panels divide over the workgroup");
    static_assert(LPT % 2 == 0, "requests are issued behind two MFMA groups");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    typedef __attribute__((address_space(3))) char* lds_cp;

    const int tid = threadIdx.x;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int wm = wave_u / WAVES_N, wn = wave_u - wm * WAVES_N;
    const int lr = lane & 31, lh = lane >> 5;
#ifdef UAVSAL_PROBE
    const int pr_bits = p.act >= 100 ? p.act - 100 : 0;
    const bool pr_mul = !(pr_bits & 2), pr_dma = !(pr_bits & 4), pr_frag = !(pr_bits & 8), pr_epi = !(pr_bits & 16);
#else
    constexpr bool pr_mul = true, pr_dma = true, pr_frag = true, pr_epi = true;
#endif
    const uavsal_tile_walk walk = xcd_tile_walk(blockIdx.x, gridDim.x, p.nblk);
    if (walk.tile >= walk.end) return;
#if K32_PRIO
    if (!(__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 1)) __builtin_amdgcn_s_setprio(2);
#endif
#ifdef UAVSAL_K32_STAMPS      // in-kernel clock: shader cycles (s_memtime) per 100 MHz tick (s_memrealtime) over the kernel
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime(), r_begin = __builtin_amdgcn_s_memrealtime();
#endif
    const int nst_all = p.Kpad / KT;
    const int ksplit = SPLIT ? p.ksplit : 1;

    // ---- request side: the stream of stages (tile by tile) still to be requested
    const int r8 = tid >> 3;
    const int lc = (tid & 7) ^ ((r8 >> 1) & 7);
    const float* a_ptr[A_IT];
    const float* b_ptr[B_IT];
    long long a_base[A_IT];
    int a_taps[A_IT];
    int rq_tile = walk.tile, rq_kt = 0, rq_nst = nst_all, it_tap = 0, it_cb = 0;
    long long tap_off = 0;
    int tap_bit = 0;
    auto setup_requests = [&](int t) {
        int s0 = 0;
        if (SPLIT) {                                     // work unit t = tile * ksplit + share
            const int sh = t % ksplit;
            t /= ksplit;
            s0 = sh * nst_all / ksplit;
            rq_nst = (sh + 1) * nst_all / ksplit - s0;
        }
        const int tile_m = t / p.tiles_n;
        const int rm0 = tile_m * BM, rn0 = (t - tile_m * p.tiles_n) * BN;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int m = rm0 + r8 + it * RPI;
            const bool ok = m < p.M;
            if (TAPS == 1) {
                a_ptr[it] = (ok ? p.a + row_off(m, p.HW, p.a_is, p.contig) * p.lda + lc * 4 : g_zero_row + lc * 4) + s0 * KT;
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
            const int nn = min(rn0 + r8 + it * RPI, p.Npad - 1);
            b_ptr[it] = reinterpret_cast<const float*>(p.w) + (size_t)nn * p.Kpad + lc * 4 + s0 * KT;
        }
        it_cb = s0 / 9; it_tap = s0 - it_cb * 9;
    };
    // moves to the next stage of the stream; false when the stream is exhausted
    auto next_request = [&]() -> bool {
        if (rq_kt == rq_nst) {
            rq_tile += walk.stride;
            rq_kt = 0;
            if (rq_tile < walk.end) setup_requests(rq_tile);
        }
        if (rq_tile >= walk.end) return false;
        ++rq_kt;
        if (TAPS == 9) {
            const int ty = (it_tap * 11) >> 5;
            tap_off = (long long)((ty - 1) * p.W + (it_tap - ty * 3 - 1)) * p.lda + it_cb * KT;
            tap_bit = it_tap;
            if (++it_tap == 9) { it_tap = 0; ++it_cb; }
        }
        return true;
    };
    auto issue_one = [&](int r, char* st) {
        if (!pr_dma) return;
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

    // ---- compute side
    const int sw = (lr >> 1) & 7;
    const unsigned lds0 = (unsigned)(size_t)(lds_cp)smem;
    const unsigned a_row = lds0 + (wm * WM * 32 + lr) * 128, b_row = lds0 + APAN + (wn * WN * 32 + lr) * 128;
    f32x16 acc[WM][WN];
    f32x4 fa[2][WM], fb[2][WN];
    // fragments of K sub-step u of the stage in ring slot `slot` -> set `buf` (WM + WN reads, not waited for)
    auto ldfrag = [&](int slot, int u, int buf) {
        if (!pr_frag) return;
        const unsigned so = slot * STAGE + (((2 * u + lh) ^ sw) << 4);
#pragma unroll
        for (int i = 0; i < WM; ++i)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[buf][i]) : "v"(a_row + so), "n"(i * 32 * 128));
#pragma unroll
        for (int j = 0; j < WN; ++j)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[buf][j]) : "v"(b_row + so), "n"(j * 32 * 128));
    };
    // set `buf` has landed when at most WM + WN younger LDS reads (the next set) are outstanding / when none is
    auto frag_wait_next = [&](int buf) {
        if (!pr_frag) return;
        asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(fa[buf][0]), "+v"(fa[buf][1]), "+v"(fb[buf][0]), "+v"(fb[buf][1])
                     : "n"(WM + WN));
    };
    auto frag_wait_all = [&](int buf) {
        if (!pr_frag) return;
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[buf][0]), "+v"(fa[buf][1]), "+v"(fb[buf][0]), "+v"(fb[buf][1]));
    };
    auto mma = [&](int buf, int i, int j) {
        if (!pr_mul) return;
        const f32x4 av = fa[buf][i], bv = fb[buf][j];
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[i][j], 0, 0, 0);
    };

    // lead-in: stage 0 -> slot 0, published; stage 1 -> slot 1; first fragments
    setup_requests(rq_tile);
    if (next_request()) {
#pragma unroll
        for (int r = 0; r < LPT; ++r) issue_one(r, smem);
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (next_request()) {
#pragma unroll
        for (int r = 0; r < LPT; ++r) issue_one(r, smem + STAGE);
    }
    ldfrag(0, 0, 0);

    int g = 0;                                           // stages done so far: stage g lives in slot g & 1
    bool req = false;                                    // requests 2 .. LPT-1 of the stage announced at the last barrier are due
    for (int unit = walk.tile; unit < walk.end; unit += walk.stride) {
        const int tile = SPLIT ? unit / ksplit : unit, share = SPLIT ? unit - tile * ksplit : 0;
        const int nst = SPLIT ? (share + 1) * nst_all / ksplit - share * nst_all / ksplit : nst_all;
        const int tile_m = tile / p.tiles_n;
        const int m0c = tile_m * BM, n0c = (tile - tile_m * p.tiles_n) * BN;
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
        const bool last_tile = unit + walk.stride >= walk.end;
        float sc[WN], bi[WN];
#pragma unroll
        for (int j = 0; j < WN && !SPLIT; ++j) {       // (scale / bias are padded to Npad; columns past it are never stored)
            const int c = min(n0c + (wn * WN + j) * 32 + lr, p.Npad - 1);
            asm volatile("global_load_dword %0, %1, off" : "=v"(sc[j]) : "v"(p.scale + c) : "memory");
            asm volatile("global_load_dword %0, %1, off" : "=v"(bi[j]) : "v"(p.bias + c) : "memory");
        }
        for (int kt = 0; kt < nst; ++kt, ++g) {
            const int slot = g & 1;
            // sub-steps 0..2: read the next sub-step's fragments, then multiply this one.  The requests of stage g + 1
            // (announced behind the previous stage's barrier, into the other slot) go out ONE behind each MFMA group:
            // a burst of them between two MFMAs stalls a wave that has its SIMD to itself for hundreds of cycles
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                ldfrag(slot, u + 1, (u + 1) & 1);
                frag_wait_next(u & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j) {
                        mma(u & 1, i, j);
                        const int r = 2 + (u * WM + i) * WN + j;
                        if (r < LPT) {
                            if (req) issue_one(r, smem + (slot ^ 1) * STAGE);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            // sub-step 3, first half
            frag_wait_all(1);
            __builtin_amdgcn_sched_barrier(0);
            mma(1, 0, 0);
            mma(1, 0, 1);
            __builtin_amdgcn_sched_barrier(0);
            // the stage's one barrier: requests of stage g + 1 landed (most of a stage old), all fragments of stage g are
            // in registers -> stage g + 1 is published and slot g & 1 is free
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            const bool more = kt + 1 < nst || !last_tile;
            if (more) ldfrag(slot ^ 1, 0, 0);
            req = next_request();                        // stage g + 2 -> slot g & 1: first two requests here, the rest above
            __builtin_amdgcn_sched_barrier(0);
            mma(1, 1, 0);
            if (req) issue_one(0, smem + slot * STAGE);
            __builtin_amdgcn_sched_barrier(0);
            mma(1, 1, 1);
            if (req) issue_one(1, smem + slot * STAGE);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- epilogue (folded BN, ReLU6 / none, optional residual), barrier-free: every wave passes its four
        // 32 x 32 accumulator blocks through its own 4 KB of LDS and stores them as 128-byte row segments.  The
        // BN scale / bias of the tile's columns were fetched at the tile's start by loads hipcc does not see (a
        // compiler-visible load left pending on some path makes it drain vmcnt(0) inside the K loop).
        if constexpr (SPLIT) {
            // (1) publish this share: raw sums -> p.kpart[share][M][Npad], device-coherent 16-byte stores
            float* stg = reinterpret_cast<float*>(smem + 2 * STAGE) + wave_u * 1024;
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) stg[((q & 3) + 8 * (q >> 2) + 4 * lh) * 32 + lr] = acc[i][j][q];
                    const int gn = n0c + (wn * WN + j) * 32 + (lane & 7) * 4;
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int row = (lane >> 3) + it * 8;
                        const int gm = m0c + (wm * WM + i) * 32 + row;
                        const f32x4 v = *reinterpret_cast<const f32x4*>(stg + row * 32 + (lane & 7) * 4);
                        float* dst = p.kpart + ((size_t)share * p.M + gm) * p.Npad + gn;
                        if (gm < p.M && gn < p.Npad)
                            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(dst), "v"(v) : "memory");
                    }
                }
            // (2) every wave's stores have left, then ONE ticket per share (MI355X_MICROARCH.md: inter-workgroup
            // visibility, the counter form); the share that draws the last ticket reduces
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (!p.sk_flag) continue;                    // the shares meet in splitk_reduce_kernel (second launch) instead
            int* ticket_lds = reinterpret_cast<int*>(smem + 2 * STAGE);
            if (tid == 0) {
                const int old = __hip_atomic_fetch_add(p.sk_flag + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int last = old == ksplit - 1;
                if (last) {
                    __hip_atomic_store(p.sk_flag + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                *ticket_lds = last;
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            const bool reducer = *ticket_lds != 0;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // (the word is staging space again)
            if (reducer) {
                // (3) shares in order 0 .. ksplit-1 (sc1 loads: served past this CU's L1), then the epilogue
                const bool twa = p.epi == UAVSAL_EPI_TWA;
                const float lo = p.act == UAVSAL_ACT_RELU6 ? 0.f : -3.0e38f;
                const float hi = p.act == UAVSAL_ACT_RELU6 ? 6.f : 3.0e38f;
                for (int it = 0; it < BM * BN / 4 / NT; it += 2) {
                    f32x4 t[2][8];
                    int gmv[2], gnv[2];
                    bool okv[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int idx = tid + (it + e) * NT;
                        const int row = idx / (BN / 4), c4 = idx - row * (BN / 4);
                        gmv[e] = m0c + row; gnv[e] = n0c + c4 * 4;
                        okv[e] = gmv[e] < p.M && gnv[e] < p.Cout;
                        const float* src = p.kpart + ((size_t)(okv[e] ? gmv[e] : 0)) * p.Npad + (okv[e] ? gnv[e] : 0);
#pragma unroll
                        for (int sh = 0; sh < 8; ++sh) {     // (no branch around a load; absent shares re-read share 0)
                            const int she = sh < ksplit ? sh : 0;
                            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(t[e][sh])
                                         : "v"(src + (size_t)she * p.M * p.Npad) : "memory");
                        }
                    }
                    asm volatile("s_waitcnt vmcnt(0)"
                                 : "+v"(t[0][0]), "+v"(t[0][1]), "+v"(t[0][2]), "+v"(t[0][3]), "+v"(t[0][4]), "+v"(t[0][5]),
                                   "+v"(t[0][6]), "+v"(t[0][7]), "+v"(t[1][0]), "+v"(t[1][1]), "+v"(t[1][2]), "+v"(t[1][3]),
                                   "+v"(t[1][4]), "+v"(t[1][5]), "+v"(t[1][6]), "+v"(t[1][7]) :: "memory");
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        if (!okv[e]) continue;
                        f32x4 v = t[e][0];
#pragma unroll
                        for (int sh = 1; sh < 8; ++sh)
                            if (sh < ksplit) v += t[e][sh];
                        const int gm = gmv[e], gn = gnv[e];
                        const long long ro = row_off(gm, p.HW, p.o_is, p.contig);
                        if (twa) {
                            const f32x4 z = v + *reinterpret_cast<const f32x4*>(p.aux + row_off(gm, p.HW, p.x_is, p.contig) * p.ldx + gn);
                            const f32x4 xt = *reinterpret_cast<const f32x4*>(p.res + row_off(gm, p.HW, p.r_is, p.contig) * p.ldr + gn);
                            const f32x4 hp = *reinterpret_cast<const f32x4*>(p.a + row_off(gm, p.HW, p.a_is, p.contig) * p.lda + gn);
                            f32x4 gt;
                            gt.x = 1.f / (1.f + expf(-z.x)); gt.y = 1.f / (1.f + expf(-z.y));
                            gt.z = 1.f / (1.f + expf(-z.z)); gt.w = 1.f / (1.f + expf(-z.w));
                            v = gt * xt + (1.f - gt) * hp;
                        } else {
                            const f32x4 s4 = p.scale ? *reinterpret_cast<const f32x4*>(p.scale + gn) : (f32x4){1.f, 1.f, 1.f, 1.f};
                            const f32x4 b4 = p.scale ? *reinterpret_cast<const f32x4*>(p.bias + gn) : (f32x4){0.f, 0.f, 0.f, 0.f};
                            v.x = __builtin_amdgcn_fmed3f(fmaf(v.x, s4.x, b4.x), lo, hi); v.y = __builtin_amdgcn_fmed3f(fmaf(v.y, s4.y, b4.y), lo, hi);
                            v.z = __builtin_amdgcn_fmed3f(fmaf(v.z, s4.z, b4.z), lo, hi); v.w = __builtin_amdgcn_fmed3f(fmaf(v.w, s4.w, b4.w), lo, hi);
                            if (p.res) v += *reinterpret_cast<const f32x4*>(p.res + row_off(gm, p.HW, p.r_is, p.contig) * p.ldr + gn);
                        }
                        *reinterpret_cast<f32x4*>(p.out + ro * p.ldc + gn) = v;
                    }
                }
            }
        } else if (pr_epi) {
            float* stg = reinterpret_cast<float*>(smem + 2 * STAGE) + wave_u * 1024;
            const float lo = p.act == UAVSAL_ACT_RELU6 ? 0.f : -3.0e38f;
            const float hi = p.act == UAVSAL_ACT_RELU6 ? 6.f : 3.0e38f;
            asm volatile("" : "+v"(sc[0]), "+v"(sc[1]), "+v"(bi[0]), "+v"(bi[1]));      // landed: older than every stage
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int r = (q & 3) + 8 * (q >> 2) + 4 * lh;
                        stg[r * 32 + lr] = __builtin_amdgcn_fmed3f(fmaf(acc[i][j][q], sc[j], bi[j]), lo, hi);
                    }
                    const int gn = n0c + (wn * WN + j) * 32 + (lane & 7) * 4;
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int row = (lane >> 3) + it * 8;
                        const int gm = m0c + (wm * WM + i) * 32 + row;
                        f32x4 v = *reinterpret_cast<const f32x4*>(stg + row * 32 + (lane & 7) * 4);
                        if (gm < p.M && gn < p.Cout) {
                            const long long ro = row_off(gm, p.HW, p.o_is, p.contig);
                            if (p.res) v += *reinterpret_cast<const f32x4*>(p.res + row_off(gm, p.HW, p.r_is, p.contig) * p.ldr + gn);
                            if (UAVSAL_STORE_OK(p.act)) *reinterpret_cast<f32x4*>(p.out + ro * p.ldc + gn) = v;
                        }
                    }
                }
        }
    }
#ifdef UAVSAL_K32_STAMPS
    if (lane == 0 && p.kpart) {
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(p.kpart);
        atomicAdd(dbg + 4, __builtin_amdgcn_s_memtime() - t_begin);
        atomicAdd(dbg + 5, 1ull);
        atomicAdd(dbg + 6, __builtin_amdgcn_s_memrealtime() - r_begin);
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------
// Few tiles, long K (the ConvTWA step: 3600 rows x 256 columns x K 2304; the projections of the 12x20 / 23x40 backbone
// maps).  What bounds such a launch is the matrix work of ONE tile on ONE CU (a 64 x 64 x 960 tile is 30 700 cycles
// of its CU's four matrix pipes whatever the workgroup looks like: splitting K over wave groups INSIDE a workgroup
// was built and measured -- 22.3 against 23.4 us -- and dropped), so the tile's K range is split over several
// workgroups = several CUs:  64 x 64 tiles, four waves, two-stage ring (32 KB: several workgroups per CU), work units
// (tile, K share).  A share publishes its 16 KB of raw sums (write-through stores) and takes a ticket on the tile's
// counter; the share that draws the last ticket adds all of them in share order -- one round of loads, 64 KB -- and
// applies the epilogue (BN / activation / residual, or the ConvTWA update).  Nobody waits, the order is fixed.
// SPLIT = false is the plain 64 x 64 instance (the usual epilogue, any epilogue kind).
template <int TAPS, bool SPLIT>
__global__ __launch_bounds__(256, 3) void conv_gemm_f32_k32s_kernel(const ConvK p) {
    constexpr int WM = 1, WN = 1, WAVES_N = 2;
    constexpr int BM = 64, BN = 64, KT = 32, NT = 256;
    constexpr int APAN = BM * 128, BPAN = BN * 128, STAGE = APAN + BPAN;
    constexpr int MAXS = 4;                              // K shares per tile at most

    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;

    const int tid = threadIdx.x;
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int wm = wave_u / WAVES_N, wn = wave_u - wm * WAVES_N;
    const int lr = lane & 31, lh = lane >> 5;
    const uavsal_tile_walk walk = xcd_tile_walk(blockIdx.x, gridDim.x, p.nblk);
    int unit = walk.tile;
    if (unit >= walk.end) return;

    const int nst_all = p.Kpad / KT;
    const int ksplit = SPLIT ? p.ksplit : 1;
    int m0 = 0, n0 = 0, nst = nst_all, tile_id = 0, share = 0;
    const int r8 = tid >> 3;
    const int lc = (tid & 7) ^ ((r8 >> 1) & 7);
    const float* a_ptr[2];
    const float* b_ptr[2];
    long long a_base[2];
    int a_taps[2];
    int it_tap = 0, it_cb = 0;
    long long tap_off = 0;
    int tap_bit = 0;
    auto setup_unit = [&](int u) {
        int s0 = 0;
        tile_id = u;
        if (SPLIT) {                                     // work unit = tile * ksplit + share
            share = u % ksplit;
            tile_id = u / ksplit;
            s0 = share * nst_all / ksplit;
            nst = (share + 1) * nst_all / ksplit - s0;
        }
        const int tile_m = tile_id / p.tiles_n;
        m0 = tile_m * BM;
        n0 = (tile_id - tile_m * p.tiles_n) * BN;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int m = m0 + r8 + it * 32;
            const bool ok = m < p.M;
            if (TAPS == 1) {
                a_ptr[it] = (ok ? p.a + row_off(m, p.HW, p.a_is, p.contig) * p.lda + (p.ngrp ? (n0 / p.ngrp) * p.a_goff : 0) + lc * 4
                                : g_zero_row + lc * 4) + s0 * KT;
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
            const int nn = min(n0 + r8 + it * 32, p.Npad - 1);
            b_ptr[it] = reinterpret_cast<const float*>(p.w) + (p.w_gs ? (size_t)(m0 / p.HW) * p.w_gs : 0) +
                        (size_t)nn * p.Kpad + lc * 4 + s0 * KT;
        }
        it_cb = s0 / 9; it_tap = s0 - it_cb * 9;
    };
    auto issue_stage = [&](char* st) {
        if (TAPS == 9) {
            const int ty = (it_tap * 11) >> 5;
            tap_off = (long long)((ty - 1) * p.W + (it_tap - ty * 3 - 1)) * p.lda + it_cb * KT;
            tap_bit = it_tap;
            if (++it_tap == 9) { it_tap = 0; ++it_cb; }
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const float* src;
            if (TAPS == 1) { src = a_ptr[it]; a_ptr[it] += KT; }
            else src = ((a_taps[it] >> tap_bit) & 1) ? p.a + a_base[it] + tap_off : g_zero16;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(st + (it * 32 + wave_u * 8) * 128), 16, 0, 0);
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            __builtin_amdgcn_global_load_lds((gptr_t)b_ptr[it], (lptr_t)(st + APAN + (it * 32 + wave_u * 8) * 128), 16, 0, 0);
            b_ptr[it] += KT;
        }
    };

    const int sw = (lr >> 1) & 7;
    const int a_row = (wm * 32 + lr) * 128, b_row = APAN + (wn * 32 + lr) * 128;
    f32x16 acc[1][1];
    auto compute = [&](const char* st) {
        f32x4 fa[2], fb[2];
        fa[0] = *reinterpret_cast<const f32x4*>(st + a_row + ((lh ^ sw) << 4));
        fb[0] = *reinterpret_cast<const f32x4*>(st + b_row + ((lh ^ sw) << 4));
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (u + 1 < 4) {
                const int so = ((2 * (u + 1) + lh) ^ sw) << 4;
                fa[(u + 1) & 1] = *reinterpret_cast<const f32x4*>(st + a_row + so);
                fb[(u + 1) & 1] = *reinterpret_cast<const f32x4*>(st + b_row + so);
            }
            const f32x4 av = fa[u & 1], bv = fb[u & 1];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[0][0], 0, 0, 0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[0][0], 0, 0, 0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[0][0], 0, 0, 0);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[0][0], 0, 0, 0);
        }
    };

    setup_unit(unit);
    issue_stage(smem);
    while (true) {
        const int m0c = m0, n0c = n0, tilec = tile_id, sharec = share, nstc = nst;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[0][0][q] = 0.f;
        for (int kt = 0; kt < nstc; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (kt + 1 < nstc) issue_stage(smem + ((kt & 1) ^ 1) * STAGE);
            compute(smem + (kt & 1) * STAGE);
        }
        const bool has_next = (unit + walk.stride) < walk.end;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // every wave is past its reads of the ring
        if (has_next) {
            unit += walk.stride;
            setup_unit(unit);
            issue_stage(smem);                           // next unit's first stage -> slot 0; the epilogue uses slot 1
        }
        if constexpr (!SPLIT) {
            UAVSAL_GEMM_EPILOGUE(1.0f, (smem + STAGE), false)
        } else {
            // (1) publish: raw sums in register order, [tile][share][16 dwords x 256 threads], device-coherent stores
            float* part = p.kpart + ((size_t)tilec * ksplit + sharec) * (BM * BN);
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const f32x4 v = {acc[0][0][4 * q4], acc[0][0][4 * q4 + 1], acc[0][0][4 * q4 + 2], acc[0][0][4 * q4 + 3]};
                float* dst = part + (q4 * 256 + tid) * 4;
                asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(dst), "v"(v) : "memory");
            }
            // (2) every wave's stores have left, then ONE ticket per share (MI355X_MICROARCH.md, inter-workgroup
            // visibility, counter form): the share that draws the last ticket reduces
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            int* ticket_lds = reinterpret_cast<int*>(smem + STAGE);
            if (tid == 0) {
                const int old = __hip_atomic_fetch_add(p.sk_flag + tilec, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int last = old == ksplit - 1;
                if (last) {
                    __hip_atomic_store(p.sk_flag + tilec, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                *ticket_lds = last;
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            const bool reducer = *ticket_lds != 0;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (reducer) {
                // (3) all shares of the tile in share order (sc1 loads, one round), then through LDS to row order
                const float* base = p.kpart + (size_t)tilec * ksplit * (BM * BN);
#pragma unroll
                for (int hq = 0; hq < 2; ++hq) {         // two rounds of 2 x ksplit loads (register budget)
                    f32x4 t[MAXS][2];
                    // (no branch around a load: a destination defined on one path only is merged by a copy BEFORE the
                    // wait below, i.e. before the data has landed; absent shares re-read share 0 and are not added)
#pragma unroll
                    for (int sh = 0; sh < MAXS; ++sh) {
                        const int she = sh < ksplit ? sh : 0;
#pragma unroll
                        for (int q = 0; q < 2; ++q)
                            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(t[sh][q])
                                         : "v"(base + (size_t)she * (BM * BN) + ((2 * hq + q) * 256 + tid) * 4) : "memory");
                    }
                    asm volatile("s_waitcnt vmcnt(0)"
                                 : "+v"(t[0][0]), "+v"(t[0][1]), "+v"(t[1][0]), "+v"(t[1][1]), "+v"(t[2][0]), "+v"(t[2][1]),
                                   "+v"(t[3][0]), "+v"(t[3][1]) :: "memory");
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        f32x4 v = t[0][q];
#pragma unroll
                        for (int sh = 1; sh < MAXS; ++sh)
                            if (sh < ksplit) v += t[sh][q];
                        const int q4 = 2 * hq + q;
                        acc[0][0][4 * q4] = v.x; acc[0][0][4 * q4 + 1] = v.y; acc[0][0][4 * q4 + 2] = v.z; acc[0][0][4 * q4 + 3] = v.w;
                    }
                }
            }
            // the reducing workgroup runs the ordinary epilogue on the summed tile (the others skip it as a whole:
            // the decision is workgroup-uniform, so are the barriers inside)
            if (reducer) UAVSAL_GEMM_EPILOGUE(1.0f, (smem + STAGE), false)
        }
        if (!has_next) break;
    }
}

// may this launch split K?  (a) the gates every in-launch reduction needs: ticket region and partial area there, an epilogue
// the reducing share carries, vector alignment, tile count inside the ticket region; (b) partial area large enough for 8 shares
// of the unpadded output (the 128 x 128 kernels; the 64 x 64 kernel sizes its own <= 4 whole-tile shares).
// The number of shares itself is uavsal_f32_k32_ksplit(tiles, stages).
bool uavsal_f32_k32_split_gates(const ConvK& k) {
    if (!k.kpart || !k.sk_flag || (k.epi != UAVSAL_EPI_AFFINE && k.epi != UAVSAL_EPI_TWA) || k.act == UAVSAL_ACT_SIGMOID) return false;
    if (k.nblk <= 0 || k.nblk >= UAVSAL_SK_TICKET_MAX) return false;
    if ((k.Cout & 3) || (k.ldc & 3) || ((size_t)k.out & 15)) return false;
    if (k.res && ((k.ldr & 3) || ((size_t)k.res & 15))) return false;
    if (k.epi == UAVSAL_EPI_TWA && ((k.ldx & 3) || (k.lda & 3) || !k.res || !k.aux || ((size_t)k.aux & 15))) return false;
    return true;
}
bool uavsal_f32_k32_split_ok(const ConvK& k) {
    return uavsal_f32_k32_split_gates(k) && 8LL * k.M * k.Npad * 4 <= k.kpart_bytes;
}

template <int WAVES_M, int WAVES_N, int MINW, bool FLAT = false>
int launch_k32(const ConvK& k0, int taps, hipStream_t stream) {
    constexpr int BM = WAVES_M * 64, BN = WAVES_N * 64, NT = WAVES_M * WAVES_N * 64;
    constexpr int SMEM = 2 * (BM + BN) * 128 + (FLAT ? 32 * BN * 4 : 0);
    ConvK k = k0;
    k.tiles_n = (k.Cout + BN - 1) / BN;
    k.nblk = ((k.M + BM - 1) / BM) * k.tiles_n;
    k.ksplit = 1;
    auto cap_of = [](auto kernel) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
        int per_cu = 0, cus = 0, dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 256;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, NT, SMEM) != hipSuccess || per_cu <= 0) per_cu = 1;
        return per_cu * cus;
    };
    if constexpr (FLAT) {
        k.ksplit = uavsal_f32_k32_split_ok(k) ? uavsal_f32_k32_ksplit(k.nblk, k.Kpad / 32) : 1;
        if (k.ksplit > 1) {          // (tile, K share) work units, reduced in the launch by the last share to arrive
            k.nblk *= k.ksplit;
            // UAVSAL_K32_FLAT_REDUCE=launch: the shares are summed by splitk_reduce_kernel instead
            static const bool by_launch = [] { const char* e = getenv("UAVSAL_K32_FLAT_REDUCE"); return e && e[0] == 'l'; }();
            if (by_launch) k.sk_flag = nullptr;
            const int cap1 = UAVSAL_PER_DEVICE(cap_of(conv_gemm_f32_k32p_kernel<WAVES_M, WAVES_N, 1, MINW, true>));
            const int cap9 = UAVSAL_PER_DEVICE(cap_of(conv_gemm_f32_k32p_kernel<WAVES_M, WAVES_N, 9, MINW, true>));
            const int cap = taps == 1 ? cap1 : cap9;
            const int grid = k.nblk < cap ? k.nblk : cap;
            if (taps == 1) {
                UAVSAL_LDS_OPTIN((conv_gemm_f32_k32p_kernel<WAVES_M, WAVES_N, 1, MINW, true>), SMEM);
                hipLaunchKernelGGL((conv_gemm_f32_k32p_kernel<WAVES_M, WAVES_N, 1, MINW, true>), dim3(grid), dim3(NT), SMEM, stream, k);
            } else {
                UAVSAL_LDS_OPTIN((conv_gemm_f32_k32p_kernel<WAVES_M, WAVES_N, 9, MINW, true>), SMEM);
                hipLaunchKernelGGL((conv_gemm_f32_k32p_kernel<WAVES_M, WAVES_N, 9, MINW, true>), dim3(grid), dim3(NT), SMEM, stream, k);
            }
            if (!k.sk_flag) { k.nblk /= k.ksplit; return launch_splitk_reduce(k, 1.0f, stream); }
        } else if (taps == 1) {
            const int cap = UAVSAL_PER_DEVICE(cap_of(conv_gemm_f32_k32p_kernel<WAVES_M, WAVES_N, 1, MINW>));
            const int grid = k.nblk < cap ? k.nblk : cap;
            UAVSAL_LDS_OPTIN((conv_gemm_f32_k32p_kernel<WAVES_M, WAVES_N, 1, MINW>), SMEM);
            hipLaunchKernelGGL((conv_gemm_f32_k32p_kernel<WAVES_M, WAVES_N, 1, MINW>), dim3(grid), dim3(NT), SMEM, stream, k);
        } else {
            const int cap = UAVSAL_PER_DEVICE(cap_of(conv_gemm_f32_k32p_kernel<WAVES_M, WAVES_N, 9, MINW>));
            const int grid = k.nblk < cap ? k.nblk : cap;
            UAVSAL_LDS_OPTIN((conv_gemm_f32_k32p_kernel<WAVES_M, WAVES_N, 9, MINW>), SMEM);
            hipLaunchKernelGGL((conv_gemm_f32_k32p_kernel<WAVES_M, WAVES_N, 9, MINW>), dim3(grid), dim3(NT), SMEM, stream, k);
        }
    } else {
        const int cap1 = UAVSAL_PER_DEVICE(cap_of(conv_gemm_f32_k32_kernel<WAVES_M, WAVES_N, 1, MINW>));
        const int cap9 = UAVSAL_PER_DEVICE(cap_of(conv_gemm_f32_k32_kernel<WAVES_M, WAVES_N, 9, MINW>));
        const int cap = taps == 1 ? cap1 : cap9;
        // Too few tiles for the chip and a long K walk (the ConvTWA step: 58 tiles x 72 stages; the 12x20 / 23x40 maps of
        // the backbone tail): K is split over up to 8 workgroups per tile, the shares meet in splitk_reduce_kernel
        // (fixed order, no atomics).  The split is a function of the shape and of fixed constants only (512 workgroup
        // slots = two per CU of a 256-CU part), not of the device the launch happens to run on.
        k.ksplit = uavsal_f32_k32_split_ok(k) ? uavsal_f32_k32_ksplit(k.nblk, k.Kpad / 32) : 1;
        k.nblk *= k.ksplit;
        const int grid = k.nblk < cap ? k.nblk : cap;
        if (taps == 1) {
            UAVSAL_LDS_OPTIN((conv_gemm_f32_k32_kernel<WAVES_M, WAVES_N, 1, MINW>), SMEM);
            hipLaunchKernelGGL((conv_gemm_f32_k32_kernel<WAVES_M, WAVES_N, 1, MINW>), dim3(grid), dim3(NT), SMEM, stream, k);
        } else {
            UAVSAL_LDS_OPTIN((conv_gemm_f32_k32_kernel<WAVES_M, WAVES_N, 9, MINW>), SMEM);
            hipLaunchKernelGGL((conv_gemm_f32_k32_kernel<WAVES_M, WAVES_N, 9, MINW>), dim3(grid), dim3(NT), SMEM, stream, k);
        }
        if (k.ksplit > 1) return launch_splitk_reduce(k, 1.0f, stream);
    }
    return uavsal_launch_status();
}

}  // namespace

// K shares per tile for `tiles` 128 x 128 tiles with `stages` 32-float K stages each (1: no split)
__attribute__((visibility("hidden"))) int uavsal_f32_k32_ksplit(long long tiles, int stages) {
    if (tiles <= 0 || tiles * 2 > 512) return 1;
    // measured on the ConvTWA step (58 tiles x 72 stages, one clip): 2 / 3 / 4 / 6 / 8 shares = 90 / 68 / 57 / 72 / 62 us
    // (kernel + reduce launch), the 64 x 64 stream-K instance it replaces 65 -- profiles/r3_gemm_k32.md
    static const int ksp_max = [] { const char* e = getenv("UAVSAL_K32_KSPLIT_MAX"); const int v = e ? atoi(e) : 4; return v < 1 ? 1 : (v > 8 ? 8 : v); }();
    int ksp = (int)(512 / tiles);
    if (ksp > ksp_max) ksp = ksp_max;
    while (ksp > 1 && stages / ksp < 6) --ksp;
    return ksp;
}

__attribute__((visibility("hidden"))) bool uavsal_f32_k32_eligible(const uavsal_conv_desc* d, int tile) {
    if (tile < 8 || tile > 11) return false;
    if (d->w_group_stride && ((tile != 8 && tile != 11) || d->taps != 1 || (((long long)d->H * d->W) & 127))) return false;
    if (d->n_group && (tile != 11 || d->taps != 1 || d->w_group_stride || d->epi != UAVSAL_EPI_AFFINE || (d->n_group & 63) ||
                       d->Cout % d->n_group || d->a_group_off < d->Cin || (d->a_group_off & 3) ||
                       (long long)d->lda < (long long)(d->Cout / d->n_group - 1) * d->a_group_off + d->Cin)) return false;
    if (d->prec != UAVSAL_PREC_F32 || d->dw_w9c || d->epi == UAVSAL_EPI_LSTM) return false;
    if (tile == 11 && d->epi == UAVSAL_EPI_TWA && ((d->ldx & 3) || (d->lda & 3) || (d->ldc & 3) || (d->Cout & 3))) return false;
    if ((d->Cin % 32) || d->Cin > UAVSAL_DWPROJ_MAX_C) return false;
    if (tile == 10) {      // the flat-pipeline kernel carries the vector affine epilogue; with K split (few tiles, long K,
                           // workspace given) the reducing share also does the ConvTWA update
        const bool al = d->act != UAVSAL_ACT_SIGMOID && !(d->ldc & 3) && !(d->Cout & 3) && uavsal_aligned16(d->out) &&
                        (!d->res || (!(d->ldr & 3) && uavsal_aligned16(d->res)));
        const long long M = (long long)d->H * d->W * d->n_img;
        const int npad = (d->Cout + 31) / 32 * 32;
        const bool split = d->sk_ws && uavsal_aligned16(d->sk_ws) && d->sk_ws_bytes - 65536 >= 8LL * M * npad * 4 &&
                           uavsal_f32_k32_ksplit(((M + 127) / 128) * ((d->Cout + 127) / 128), d->taps * d->Cin / 32) > 1;
        if (split) {
            if (!al || (d->epi != UAVSAL_EPI_AFFINE && d->epi != UAVSAL_EPI_TWA)) return false;
            if (d->epi == UAVSAL_EPI_TWA && ((d->ldx & 3) || (d->lda & 3) || !d->res || !d->aux || !uavsal_aligned16(d->aux))) return false;
        } else if (!(al && d->epi == UAVSAL_EPI_AFFINE && d->scale && d->bias)) {
            return false;
        }
    }
    return d->taps == 1 || d->taps == 9;
}

namespace {
// 64 x 64 tiles; K shares per tile (<= 4) when the workspace is there and the tiles alone leave most of the chip idle:
// shares = the count that brings the work units to about two per CU of a 256-CU part (a function of the shape only)
int launch_k32s(const ConvK& k0, int taps, hipStream_t stream) {
    constexpr int SMEM = 2 * (64 + 64) * 128;
    ConvK k = k0;
    k.tiles_n = (k.Cout + 63) / 64;
    k.nblk = ((k.M + 63) / 64) * k.tiles_n;
    const int stages = k.Kpad / 32;
    int ksp = 1;
    if (uavsal_f32_k32_split_gates(k)) {
        ksp = (int)(1024 / (k.nblk > 0 ? k.nblk : 1));
        if (ksp > 4) ksp = 4;
        while (ksp > 1 && stages / ksp < 4) --ksp;
        if (ksp < 1) ksp = 1;
        if ((long long)k.nblk * ksp * 64 * 64 * 4 > k.kpart_bytes) ksp = 1;
    }
    k.ksplit = ksp;
    auto cap_of = [](auto kernel) {
        int per_cu = 0, cus = 0, dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 256;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, SMEM) != hipSuccess || per_cu <= 0) per_cu = 1;
        return per_cu * cus;
    };
    if (ksp > 1) {
        k.nblk *= ksp;
        const int cap1 = UAVSAL_PER_DEVICE(cap_of(conv_gemm_f32_k32s_kernel<1, true>));
        const int cap9 = UAVSAL_PER_DEVICE(cap_of(conv_gemm_f32_k32s_kernel<9, true>));
        const int cap = taps == 1 ? cap1 : cap9;
        const int grid = k.nblk < cap ? k.nblk : cap;
        if (taps == 1) hipLaunchKernelGGL((conv_gemm_f32_k32s_kernel<1, true>), dim3(grid), dim3(256), SMEM, stream, k);
        else hipLaunchKernelGGL((conv_gemm_f32_k32s_kernel<9, true>), dim3(grid), dim3(256), SMEM, stream, k);
    } else {
        const int cap1 = UAVSAL_PER_DEVICE(cap_of(conv_gemm_f32_k32s_kernel<1, false>));
        const int cap9 = UAVSAL_PER_DEVICE(cap_of(conv_gemm_f32_k32s_kernel<9, false>));
        const int cap = taps == 1 ? cap1 : cap9;
        const int grid = k.nblk < cap ? k.nblk : cap;
        if (taps == 1) hipLaunchKernelGGL((conv_gemm_f32_k32s_kernel<1, false>), dim3(grid), dim3(256), SMEM, stream, k);
        else hipLaunchKernelGGL((conv_gemm_f32_k32s_kernel<9, false>), dim3(grid), dim3(256), SMEM, stream, k);
    }
    return uavsal_launch_status();
}
}  // namespace

__attribute__((visibility("hidden"))) int uavsal_launch_f32_k32(const uavsal_gemm::ConvK& k, int taps, int tile,
                                                                 hipStream_t stream) {
    if (tile == 11) return launch_k32s(k, taps, stream);            // 64 x 64, K shares over workgroups, reduced in the launch
    if (tile == 10) return launch_k32<2, 2, 2, true>(k, taps, stream);    // 128 x 128, flat pipeline, 80 KB: two per CU
    if (tile == 9) return launch_k32<4, 2, 2>(k, taps, stream);     // 256 x 128 on 8 waves, one workgroup per CU
    return launch_k32<2, 2, 2>(k, taps, stream);                    // 128 x 128 on 4 waves, two per CU
}

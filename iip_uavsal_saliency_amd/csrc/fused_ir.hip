// Fused inverted-residual block for the small-channel, HBM-bound head of the backbone:
//   out = bn3(pw_linear( relu6(bn2(dw3x3_s( relu6(bn1(pw_expand(x))) ))) )) [+ x]
// i.e. torchvision InvertedResidual / dwBlock (reference model.py:74-103) in ONE launch.  The 6x expanded
// tensors E and D never reach HBM: at 360x640 the unfused features.2 block moves 482 MB (E alone is 177 MB
// written + 177 MB read) for 40 MB of block input + output.
//
// One 256-thread workgroup produces an 8x8 patch of output pixels of one image.  The input patch with its
// halo ((8-1)*S+3 squared pixels) is staged once in LDS; the hidden channels are processed in chunks of HC:
//   expand : E[halo pixel][c] = x . W1 on v_mfma_f32_16x16x4_f32 (exact fp32, the vector-FMA rate, but the
//            weights are B fragments held in VGPRs: the first version of this kernel used scalar-operand
//            v_fmac and spent its time waiting for s_load -- it was slower than the three separate launches).
//            A = x from LDS (one ds_read_b32 per MFMA), 16-pixel row tiles dealt round-robin to the 4 waves;
//            relu6(bn1) and EXACT ZERO outside the image (the depthwise conv pads E, not x) -> LDS, channel-major;
//   dw     : lane = (channel, pixel group): nine taps from LDS, tap weights / BN in VGPRs, relu6(bn2) -> D (LDS);
//   project: wave w owns output pixels 16w..16w+15: OUT += D . W2 on the same MFMA, accumulators live across
//            the chunks (4 VGPRs per 16 output channels);
// epilogue : bn3, optional residual (the block input at the same pixel), 64-byte stores.
// MFMA lane maps (cdna_hip_programming.md): A[l & 15][k = l >> 4], B[k = l >> 4][l & 15],
// D[row = 4 * (l >> 4) + reg][col = l & 15].
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

template <int CIN, int HID, int COUT, int S, bool EXPAND, int HC>
__global__ __launch_bounds__(256) void fused_ir_kernel(const FusedK p) {
    constexpr int TY = 8, TX = 8;
    constexpr int IH = (TY - 1) * S + 3, IW = (TX - 1) * S + 3, NIN = IH * IW;
    constexpr int NRT = (NIN + 15) / 16, NINP = NRT * 16;       // halo pixels in row tiles of 16
    constexpr int ESTR = NINP + 1;                              // 17 mod 32: lanes = 16 channels hit 16 banks
    constexpr int DSTR = 81;
    constexpr int NCH = HID / HC;
    constexpr int NCTE = HC / 16;                               // column tiles of the expand GEMM
    constexpr int NCT = (COUT + 15) / 16;                       // column tiles of the projection GEMM
    constexpr int KE = CIN / 4, KP = HC / 4;                    // MFMA k steps
    constexpr int NJOB = HC / 16, NPART = 4 / NJOB;             // depthwise: channel blocks x pixel parts over the waves
    constexpr int PXL = 64 / (4 * NPART);                       // output pixels per lane in the depthwise phase
    static_assert(HID % HC == 0 && HC % 16 == 0 && CIN % 4 == 0 && COUT % 8 == 0, "channel blocking");
    static_assert(EXPAND || (HID == CIN && NCH == 1), "without an expand conv the hidden tensor is the input");

    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* xs = lds;                                            // [KE][NINP][4]   (EXPAND only)
    float* es = lds + (EXPAND ? CIN * NINP : 0);                // [HC][ESTR]
    float* ds = es + HC * ESTR;                                 // [HC][DSTR]

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

    // ---- stage the input patch (zero outside the image and in the pad rows) -------------------------
    for (int idx = tid; idx < NINP * KE; idx += 256) {
        const int pix = idx / KE, k4 = idx - pix * KE;
        const int iy = iy0 + pix / IW, ix = ix0 + pix % IW;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (pix < NIN && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
            v = *reinterpret_cast<const f32x4*>(inb + ((size_t)iy * p.W + ix) * p.ldi + k4 * 4);
        if (EXPAND) {
            *reinterpret_cast<f32x4*>(xs + ((size_t)k4 * NINP + pix) * 4) = v;
        } else {                                   // the input IS the hidden tensor: channel-major scalars
            es[(k4 * 4 + 0) * ESTR + pix] = v.x; es[(k4 * 4 + 1) * ESTR + pix] = v.y;
            es[(k4 * 4 + 2) * ESTR + pix] = v.z; es[(k4 * 4 + 3) * ESTR + pix] = v.w;
        }
    }

    f32x4 acc_o[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) acc_o[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int ch = 0; ch < NCH; ++ch) {
        const int c0 = ch * HC;
        // ---- this chunk's weights, as MFMA B fragments / per-lane constants (global loads, L2-resident) ----
        float w1f[EXPAND ? NCTE : 1][EXPAND ? KE : 1], s1f[EXPAND ? NCTE : 1], b1f[EXPAND ? NCTE : 1];
        if (EXPAND) {
#pragma unroll
            for (int ct = 0; ct < NCTE; ++ct) {
                const int c = c0 + ct * 16 + l15;
#pragma unroll
                for (int s = 0; s < KE; ++s) w1f[ct][s] = p.w1[(size_t)(4 * s + lq) * HID + c];
                s1f[ct] = p.s1[c]; b1f[ct] = p.b1[c];
            }
        }
        float w2f[NCT][KP];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const int co = ct * 16 + l15;
#pragma unroll
            for (int s = 0; s < KP; ++s) w2f[ct][s] = co < COUT ? p.w2[(size_t)(c0 + 4 * s + lq) * COUT + co] : 0.f;
        }
        // depthwise: lane = channel (wave % NJOB) * 16 + l15 of the chunk, pixel part (wave / NJOB, lq)
        const bool dw_on = wave < NJOB * NPART;
        const int dj = wave % NJOB, dpart = wave / NJOB;
        const int dcl = dj * 16 + l15;                          // channel inside the chunk
        float wdf[9], sdf, bdf;
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) wdf[tp] = p.wd[tp * HID + c0 + dcl];
        sdf = p.sd[c0 + dcl]; bdf = p.bd[c0 + dcl];

        __syncthreads();          // input patch staged (ch == 0) / previous chunk's projection done with D
        if (EXPAND) {
            for (int rt = wave; rt < NRT; rt += 4) {
                float a[KE];
#pragma unroll
                for (int s = 0; s < KE; ++s) a[s] = xs[((size_t)s * NINP + rt * 16 + l15) * 4 + lq];
                float inside[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int pix = rt * 16 + 4 * lq + i;
                    const int iy = iy0 + pix / IW, ix = ix0 + pix % IW;
                    inside[i] = (pix < NIN && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) ? 1.f : 0.f;
                }
#pragma unroll
                for (int ct = 0; ct < NCTE; ++ct) {
                    f32x4 e = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < KE; ++s) e = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], w1f[ct][s], e, 0, 0, 0);
                    float* dst = es + (size_t)(ct * 16 + l15) * ESTR + rt * 16 + 4 * lq;
#pragma unroll
                    for (int i = 0; i < 4; ++i)     // zero padding of the depthwise conv: E = 0 outside the image
                        dst[i] = fminf(fmaxf(fmaf(e[i], s1f[ct], b1f[ct]), 0.f), 6.f) * inside[i];
                }
            }
            __syncthreads();
        }
        // ---- depthwise 3x3 (stride S) + BN + ReLU6 -> D ------------------------------------------------
        if (dw_on) {
            const float* e = es + (size_t)dcl * ESTR;
#pragma unroll 2
            for (int q = 0; q < PXL; ++q) {
                const int px = (dpart * 4 + lq) * PXL + q;          // output pixel of the 8x8 patch
                const float* ep = e + ((px >> 3) * S) * IW + (px & 7) * S;
                float d = 0.f;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) d = fmaf(ep[ky * IW + kx], wdf[ky * 3 + kx], d);
                ds[dcl * DSTR + px] = fminf(fmaxf(fmaf(d, sdf, bdf), 0.f), 6.f);
            }
        }
        __syncthreads();          // D visible; es free for the next chunk's expand
        // ---- projection: output pixels 16*wave .. +15, all output channels, K = this chunk's HC channels ----
#pragma unroll
        for (int s = 0; s < KP; ++s) {
            const float a = ds[(4 * s + lq) * DSTR + wave * 16 + l15];
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) acc_o[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, w2f[ct][s], acc_o[ct], 0, 0, 0);
        }
    }

    // ---- epilogue: BN (linear), residual, store: lane holds pixels 16*wave + 4*lq + i, channel ct*16 + l15 ----
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        const int co = ct * 16 + l15;
        if (co >= COUT) continue;
        const float sc = p.s2[co], bi = p.b2[co];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int px = wave * 16 + 4 * lq + i;
            const int gy = oy0 + (px >> 3), gx = ox0 + (px & 7);
            if (gy >= p.Ho || gx >= p.Wo) continue;
            const size_t opix = ((size_t)n * p.Ho + gy) * p.Wo + gx;
            float v = fmaf(acc_o[ct][i], sc, bi);
            if (p.res) v += p.res[opix * p.ldr + co];
            p.out[opix * p.ldo + co] = v;
        }
    }
}

template <int CIN, int HID, int COUT, int S, bool EXPAND, int HC>
int launch_fused(const uavsal_fused_ir_desc* d, hipStream_t s) {
    constexpr int IH = 7 * S + 3, NIN = IH * IH, NINP = (NIN + 15) / 16 * 16;
    constexpr size_t SMEM = 4 * ((EXPAND ? CIN * NINP : 0) + (size_t)HC * (NINP + 1) + (size_t)HC * 81);
    FusedK k;
    k.in = d->in; k.w1 = d->w1; k.s1 = d->scale1; k.b1 = d->bias1;
    k.wd = d->wd; k.sd = d->scale_d; k.bd = d->bias_d;
    k.w2 = d->w2; k.s2 = d->scale2; k.b2 = d->bias2;
    k.res = d->res; k.out = d->out;
    k.ldi = d->ldi; k.ldr = d->ldr; k.ldo = d->ldo; k.H = d->H; k.W = d->W;
    k.Ho = (d->H - 1) / S + 1; k.Wo = (d->W - 1) / S + 1;
    k.tiles_x = (k.Wo + 7) / 8; k.tiles_y = (k.Ho + 7) / 8;
    const long long nblk = (long long)d->n_img * k.tiles_y * k.tiles_x;
    if (nblk > 0x7fffffffLL) return UAVSAL_ESHAPE;
    hipLaunchKernelGGL((fused_ir_kernel<CIN, HID, COUT, S, EXPAND, HC>), dim3((unsigned)nblk), dim3(256), SMEM, s, k);
    return uavsal_launch_status();
}

// the channel / stride combinations that exist as instances: MobileNetV2 features[1..7] (model_feature.py:62-66)
int dispatch(const uavsal_fused_ir_desc* d, hipStream_t s, bool launch) {
#define UAVSAL_FUSED_CASE(CIN, HID, COUT, S, EXP, HCC)                                                   \
    if (d->Cin == CIN && d->hidden == HID && d->Cout == COUT && d->stride == S && (d->w1 != nullptr) == EXP) \
        return launch ? launch_fused<CIN, HID, COUT, S, EXP, HCC>(d, s) : 1;
#ifndef UAVSAL_FUSED_HC_A
#define UAVSAL_FUSED_HC_A 48      /* hidden chunk of the stride-1 blocks */
#endif
#ifndef UAVSAL_FUSED_HC_B
#define UAVSAL_FUSED_HC_B 32      /* ... of the stride-2 blocks with hidden % 32 == 0 */
#endif
#ifndef UAVSAL_FUSED_HC_C
#define UAVSAL_FUSED_HC_C 16      /* ... of features.4 (hidden 144: 16 or 48) */
#endif
    UAVSAL_FUSED_CASE(32, 32, 16, 1, false, 32)     // features.1  (t = 1: no expand conv)
    UAVSAL_FUSED_CASE(16, 96, 24, 2, true, UAVSAL_FUSED_HC_B)      // features.2
    UAVSAL_FUSED_CASE(24, 144, 24, 1, true, UAVSAL_FUSED_HC_A)     // features.3
    UAVSAL_FUSED_CASE(24, 144, 32, 2, true, UAVSAL_FUSED_HC_C)     // features.4
    UAVSAL_FUSED_CASE(32, 192, 32, 1, true, UAVSAL_FUSED_HC_A)     // features.5, features.6
    UAVSAL_FUSED_CASE(32, 192, 64, 2, true, UAVSAL_FUSED_HC_B)     // features.7
#undef UAVSAL_FUSED_CASE
    return launch ? UAVSAL_ESHAPE : 0;
}

}  // namespace

extern "C" int uavsal_fused_ir_supported(const uavsal_fused_ir_desc* d) {
    if (!d) return 0;
    return dispatch(d, nullptr, false);
}

extern "C" int uavsal_fused_ir(const uavsal_fused_ir_desc* d, uavsal_stream_t stream) {
    if (!d || !d->in || !d->wd || !d->scale_d || !d->bias_d || !d->w2 || !d->scale2 || !d->bias2 || !d->out) return UAVSAL_EINVAL;
    if (d->w1 && (!d->scale1 || !d->bias1)) return UAVSAL_EINVAL;
    if (d->n_img <= 0 || d->H <= 0 || d->W <= 0) return UAVSAL_EINVAL;
    if ((d->ldi & 3) || d->ldi < d->Cin || d->ldo < d->Cout || !uavsal_aligned16(d->in)) return UAVSAL_EALIGN;
    if (d->res && (d->ldr < d->Cout || d->stride != 1 || d->Cin != d->Cout)) return UAVSAL_ESHAPE;
    return dispatch(d, (hipStream_t)stream, true);
}

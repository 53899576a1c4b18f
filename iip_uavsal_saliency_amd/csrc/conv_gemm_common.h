// Shared between the GEMM translation units (conv_gemm.hip, conv_gemm_k32.hip): the kernel argument block,
// the epilogue every GEMM kernel expands, the zero pages out-of-range lanes fetch, probe switches.
#pragma once
#include <stdlib.h>
#include <type_traits>
#include "common.h"

namespace uavsal_gemm {

struct ConvK {
    const float* a;
    const char* w;
    const float* scale;
    const float* bias;
    float* out;
    const float* res;
    const float* aux;
    float* out2;
    const float* dw_w; const float* dw_s; const float* dw_b;   // fused depthwise producer (or null)
    int dw_stride, Hin, Win;
    long long a_is, o_is, r_is, x_is;
    int lda, ldc, ldr, ldx, ld2;
    int M, HW, H, W, Cin, Cout, Kpad, Npad, ktiles, act, epi;
    int tiles_n, nblk, contig;
    float* sk_part;      // stream-K: one BM x BN fp32 partial per workgroup ...
    int* sk_flag;        // ... and its "published" flag (0 at launch, reset by the consumer)
    int* err;            // device error word (UAVSAL_ERR_*)
    int sk_spin, sk_drop;
    int ksplit;                // K split (dwproj_kernel, the 64x64 3x3 register-staged tile): workgroups per output tile
                               // along K; 1: the tile's epilogue runs in place
    float* kpart;              // ... their raw partial sums [ksplit][M][Npad] (splitk_reduce_kernel finishes them)
    long long kpart_bytes;     // ... and the room there is for them
    const _Float16* a_sp;      // pre-split A operand (split shadow, uavsal_hip.h) or null
    _Float16* out_sp;          // optional split shadow of the output
    int ldas, ldos;            // their row strides in halves
    long long w_gs;            // per-image weights (uavsal_conv_desc.w_group_stride, floats; 0 = none): HW % 128 == 0
    int ngrp, a_goff;          // output-channel groups with their own A columns (uavsal_conv_desc.n_group / a_group_off; 0 = none)
};

}  // namespace uavsal_gemm

// conv_gemm_k32.hip: fp32 LDS-DMA GEMM with 128-byte (32-float) K stages; tile = 8 (128 x 128) or 9 (256 x 128)
int uavsal_launch_f32_k32(const uavsal_gemm::ConvK& k, int taps, int tile, hipStream_t stream);
bool uavsal_f32_k32_eligible(const uavsal_conv_desc* d, int tile);
int uavsal_f32_k32_ksplit(long long tiles, int stages);
// dwproj.hip: depthwise 3x3 + BN + ReLU6 -> 1x1 projection in one launch (LDS halo tile), fp32 / split-fp16
int uavsal_launch_dwproj(const uavsal_gemm::ConvK& k, int prec, hipStream_t stream);

namespace {
using uavsal_gemm::ConvK;

__device__ __forceinline__ long long row_off(int m, int HW, long long img_stride, int contig) {
    if (contig) return m;
    const int img = m / HW;
    return (long long)img * img_stride + (m - img * HW);
}

// F16X3 pre-scales: activations by 2^4 while staging (saturating at the fp16 range: hi and lo
// each stop at 65504, i.e. |x| beyond ~8188 clips), weights by 2^6 on the host; the accumulator
// is scaled back by 2^-10.
// Powers of two, so nothing is rounded by the scaling itself; it only keeps the low halves
// of the split away from the fp16 subnormal range.
#define F16X3_A_SCALE 16.0f
#define F16X3_ACC_SCALE (1.0f / 1024.0f)

template <int PREC>
__device__ __forceinline__ unsigned pack2(float a, float b) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 v = {a, b};
    if (PREC == UAVSAL_PREC_F16X3) {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        h2 r = __builtin_convertvector(v, h2);
        return __builtin_bit_cast(unsigned, r);
    } else {
        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
        bf2 r = __builtin_convertvector(v, bf2);
        return __builtin_bit_cast(unsigned, r);
    }
}

// value of the 16-bit rounding of x, as fp32 (same conversion as pack2)
template <int PREC>
__device__ __forceinline__ float hi_as_f32(float x) {
    if (PREC == UAVSAL_PREC_F16X3) { _Float16 h = (_Float16)x; return (float)h; }
    __bf16 h = (__bf16)x;
    return (float)h;
}

template <int PREC>
__device__ __forceinline__ f32x4 prescale(f32x4 x) {
    if (PREC == UAVSAL_PREC_F16X3) {
        x = x * F16X3_A_SCALE;
        x.x = fminf(fmaxf(x.x, -65504.f), 65504.f); x.y = fminf(fmaxf(x.y, -65504.f), 65504.f);
        x.z = fminf(fmaxf(x.z, -65504.f), 65504.f); x.w = fminf(fmaxf(x.w, -65504.f), 65504.f);
    }
    return x;
}

// ---- epilogue ---------------------------------------------------------------------------
// The accumulators leave through one of two paths.  The fast path covers every launch but the
// ConvTWA step and the final sigmoid: dense images (img stride == H*W), act in {none, ReLU6}.
// Its per-element cost is what bounds the short-K layers (K = 16..256: the epilogue used to
// take as long as the MFMAs), so it is kept to: fma, v_med3 clamp, optional residual load+add,
// one store with a wave-uniform base + 32-bit lane offset.  Written as a macro expanded inside
// each kernel: handing the accumulator array to a function demotes it to scratch memory.
// (the two workgroup barriers of the staged path: a translation unit whose kernels keep LDS-DMA requests in flight
// across the epilogue defines UAVSAL_EPI_BARRIER as an asm `s_waitcnt lgkmcnt(0); s_barrier` before including this
// header -- __syncthreads() makes hipcc drain vmcnt(0) first)
#ifndef UAVSAL_EPI_BARRIER
#define UAVSAL_EPI_BARRIER() __syncthreads()
#endif
#define UAVSAL_GEMM_EPILOGUE(ACC_SCALE, STG, SPLIT_OUT)                                                       \
    {                                                                                                \
        /* the vector ConvTWA update is only compiled into the small-tile kernels (register budget) */ \
        const bool twa_ = (WM * WN == 1) && p.epi == UAVSAL_EPI_TWA;                                 \
        const bool lstm_ = (WM * WN == 1) && p.epi == UAVSAL_EPI_LSTM;                               \
        const bool vec_ = (p.epi == UAVSAL_EPI_AFFINE || twa_ || lstm_) &&                           \
                          p.act != UAVSAL_ACT_SIGMOID && !(p.ldc & 3) && !(p.Cout & 3) &&            \
                          !((size_t)p.out & 15) &&                                                   \
                          (!p.res || (!(p.ldr & 3) && !((size_t)p.res & 15))) &&                     \
                          (!twa_ || (!(p.ldx & 3) && !((size_t)p.aux & 15) && !(p.lda & 3)));        \
        float sc[WN], bi[WN];                                                                        \
        int col[WN];                                                                                 \
        bool cok[WN];                                                                                \
        _Pragma("unroll") for (int j = 0; j < WN; ++j) {                                             \
            col[j] = (wn * WN + j) * 32 + lr;                                                        \
            cok[j] = n0c + col[j] < p.Cout;                                                          \
            const bool okn = (p.scale != nullptr) && cok[j];                                         \
            sc[j] = (okn ? p.scale[n0c + col[j]] : 1.f) * (ACC_SCALE);                               \
            bi[j] = okn ? p.bias[n0c + col[j]] : 0.f;                                                \
        }                                                                                            \
        if (vec_) {                                                                                  \
            /* 32-row blocks go through LDS so that every lane stores 16 B and a wave covers whole  \
               rows: 4x fewer store instructions (the store queue, not HBM, bounds short-K layers) */ \
            const float lo = p.act == UAVSAL_ACT_RELU6 ? 0.f : -3.0e38f;                             \
            const float hi = p.act == UAVSAL_ACT_RELU6 ? 6.f : 3.0e38f;                              \
            float* stg = reinterpret_cast<float*>(STG);                                              \
            /* 32-row block pp = w * WM + i belongs to the waves with wm == w: `i` stays a compile-time \
               index into the accumulators (a run-time one would demote them to scratch) */           \
            _Pragma("unroll") for (int i = 0; i < WM; ++i)                                           \
            _Pragma("unroll") for (int w = 0; w < BM / 32 / WM; ++w) {                               \
                const int pp = w * WM + i;                                                           \
                if (wm == w) {                                                                       \
                    _Pragma("unroll") for (int g = 0; g < 16; ++g) {                                 \
                        const int r = (g & 3) + 8 * (g >> 2) + 4 * lh;                               \
                        _Pragma("unroll") for (int j = 0; j < WN; ++j)                               \
                            stg[r * BN + col[j]] = (twa_ || lstm_) ? acc[i][j][g] * (ACC_SCALE) :    \
                                __builtin_amdgcn_fmed3f(fmaf(acc[i][j][g], sc[j], bi[j]), lo, hi);   \
                    }                                                                                \
                }                                                                                    \
                UAVSAL_EPI_BARRIER();                                                                   \
                _Pragma("unroll") for (int it = 0; it < (32 * BN / 4 + NT - 1) / NT; ++it) {         \
                    const int idx = tid + it * NT;                                                   \
                    const int row = idx / (BN / 4), c4 = idx - row * (BN / 4);                       \
                    const int gm = m0c + pp * 32 + row, gn = n0c + c4 * 4;                           \
                    if (row < 32 && gm < p.M && gn < p.Cout) {                                       \
                        f32x4 v = *reinterpret_cast<const f32x4*>(stg + row * BN + c4 * 4);          \
                        const long long oo = row_off(gm, p.HW, p.o_is, p.contig) * p.ldc + gn;       \
                        if (lstm_) {                                                                 \
                            const f32x4 z = v + *reinterpret_cast<const f32x4*>(                     \
                                p.aux + row_off(gm, p.HW, p.x_is, p.contig) * p.ldx + gn);           \
                            const int ch = gn >> 2;                                                  \
                            const float cp = p.res[row_off(gm, p.HW, p.r_is, p.contig) * p.ldr + ch]; \
                            const float ig = 1.f / (1.f + expf(-z.x)), fg = 1.f / (1.f + expf(-z.y)); \
                            const float og = 1.f / (1.f + expf(-z.z)), gg = tanhf(z.w);              \
                            const float cn = fg * cp + ig * gg;                                      \
                            const long long ho = row_off(gm, p.HW, p.o_is, p.contig);                \
                            p.out[ho * p.ldc + ch] = og * tanhf(cn);                                 \
                            p.out2[ho * p.ld2 + ch] = cn;                                            \
                            continue;                                                                \
                        }                                                                            \
                        if (twa_) {                                                                  \
                            const f32x4 z = v + *reinterpret_cast<const f32x4*>(                     \
                                p.aux + row_off(gm, p.HW, p.x_is, p.contig) * p.ldx + gn);           \
                            const f32x4 xt = *reinterpret_cast<const f32x4*>(                        \
                                p.res + row_off(gm, p.HW, p.r_is, p.contig) * p.ldr + gn);           \
                            const f32x4 hp = *reinterpret_cast<const f32x4*>(                        \
                                p.a + row_off(gm, p.HW, p.a_is, p.contig) * p.lda + gn);             \
                            f32x4 gt;                                                                \
                            gt.x = 1.f / (1.f + expf(-z.x)); gt.y = 1.f / (1.f + expf(-z.y));        \
                            gt.z = 1.f / (1.f + expf(-z.z)); gt.w = 1.f / (1.f + expf(-z.w));        \
                            v = gt * xt + (1.f - gt) * hp;                                           \
                        } else if (p.res) {                                                          \
                            v += *reinterpret_cast<const f32x4*>(                                    \
                                p.res + row_off(gm, p.HW, p.r_is, p.contig) * p.ldr + gn);           \
                        }                                                                            \
                        if (UAVSAL_STORE_OK(p.act)) *reinterpret_cast<f32x4*>(p.out + oo) = v;               \
                        if ((SPLIT_OUT) && p.out_sp)  /* split shadow for the GEMMs that consume it */  \
                            uavsal_store_split4(p.out_sp + row_off(gm, p.HW, p.o_is, p.contig) * p.ldos, gn, v); \
                    }                                                                                \
                }                                                                                    \
                UAVSAL_EPI_BARRIER();                                                                   \
            }                                                                                        \
        } else if (WM * WN <= 4) {   /* scalar path; not built for the 256 x 256 tile (host checks) */ \
            _Pragma("unroll") for (int i = 0; i < WM; ++i) {                                         \
                _Pragma("unroll") for (int g = 0; g < 16; ++g) {                                     \
                    const int m = m0c + (wm * WM + i) * 32 + (g & 3) + 8 * (g >> 2) + 4 * lh;        \
                    if (m >= p.M) continue;                                                          \
                    const long long oo = row_off(m, p.HW, p.o_is, p.contig) * p.ldc;                 \
                    long long ro = 0, xo = 0, ao = 0;                                                \
                    if (p.res) ro = row_off(m, p.HW, p.r_is, p.contig) * p.ldr;                      \
                    if (p.epi == UAVSAL_EPI_TWA) {                                                   \
                        xo = row_off(m, p.HW, p.x_is, p.contig) * p.ldx;                             \
                        ao = row_off(m, p.HW, p.a_is, p.contig) * p.lda;                             \
                    }                                                                                \
                    _Pragma("unroll") for (int j = 0; j < WN; ++j) {                                 \
                        const int n = n0c + col[j];                                                  \
                        if (!cok[j]) continue;                                                       \
                        float vv;                                                                    \
                        if (p.epi == UAVSAL_EPI_TWA) {                                               \
                            const float z = acc[i][j][g] * (ACC_SCALE) + p.aux[xo + n];              \
                            const float gate = 1.f / (1.f + expf(-z));                               \
                            const float xt = p.res[ro + n];                                          \
                            const float hp = p.a[ao + n];                                            \
                            vv = gate * xt + (1.f - gate) * hp;                                      \
                        } else {                                                                     \
                            vv = apply_act(fmaf(acc[i][j][g], sc[j], bi[j]), p.act);                 \
                            if (p.res) vv += p.res[ro + n];                                          \
                        }                                                                            \
                        p.out[oo + n] = vv;                                                          \
                    }                                                                                \
                }                                                                                    \
            }                                                                                        \
        }                                                                                            \
    }

// 16 bytes of zeros: what out-of-range lanes (M / K tails, 3x3 zero padding) fetch instead of
// being masked off
__device__ __attribute__((aligned(16))) float g_zero16[4];
// zeros for a whole K walk (dwproj_kernel: out-of-image halo pixels advance through it like real rows do)
#define UAVSAL_DWPROJ_MAX_C 4096
__device__ __attribute__((aligned(16))) float g_zero_row[UAVSAL_DWPROJ_MAX_C + 16];

#ifndef UAVSAL_SK_PREFETCH
#define UAVSAL_SK_PREFETCH 1
#endif
// tools/gemm_probe2.py builds with -DUAVSAL_PROBE: act codes >= 100 then time the kernel without its output store
// (100), without its MFMA loop (101) or without its DMA requests (102); the product build has none of this
#ifdef UAVSAL_PROBE
#define UAVSAL_STORE_OK(act) ((act) < 100)
#else
#define UAVSAL_STORE_OK(act) true
#endif
#ifndef UAVSAL_SK_ACQUIRE
#define UAVSAL_SK_ACQUIRE 1
#endif
#ifndef UAVSAL_H16_STAGGER
#define UAVSAL_H16_STAGGER 1
#endif
#ifndef UAVSAL_GEMM_PREFETCH
#define UAVSAL_GEMM_PREFETCH 2
#endif

// Second launch of a K-split GEMM (dwproj_kernel's narrow instance, the 64x64 3x3 tile of the ConvTWA step): sums the
// shares in a fixed order, then the epilogue -- BN, activation, residual, or the ConvTWA update (model_convlstm.py:
// 276-292: gate = sigmoid(sum + W_x x_t), h_t = gate x_t + (1 - gate) h_{t-1}) -- and the split shadow.
// One thread per (row, 4 output channels).
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const ConvK p, float acc_scale) {
    const int groups = (p.Cout + 3) >> 2;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long long)p.M * groups) return;
    const int m = (int)(idx / groups), gn = (int)(idx - (long long)m * groups) * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < p.ksplit; ++k) v += *reinterpret_cast<const f32x4*>(p.kpart + ((size_t)k * p.M + m) * p.Npad + gn);
    const int img = m / p.HW, pix = m - img * p.HW;
    float* o = p.out + ((long long)img * p.o_is + pix) * p.ldc + gn;
    const float* rs = p.res ? p.res + ((long long)img * p.r_is + pix) * p.ldr + gn : nullptr;
    const bool twa = p.epi == UAVSAL_EPI_TWA;
    const float* ax = twa ? p.aux + ((long long)img * p.x_is + pix) * p.ldx + gn : nullptr;
    const float* hp = twa ? p.a + ((long long)img * p.a_is + pix) * p.lda + gn : nullptr;
    const bool vec = gn + 3 < p.Cout && !(p.ldc & 3) && !((size_t)p.out & 15) && (!p.res || (!(p.ldr & 3) && !((size_t)p.res & 15)));
    f32x4 r;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const bool okc = gn + c < p.Cout;
        if (twa) {
            const float z = v[c] * acc_scale + (okc ? ax[c] : 0.f);
            const float gate = 1.f / (1.f + expf(-z));
            r[c] = okc ? gate * rs[c] + (1.f - gate) * hp[c] : 0.f;
        } else {
            const float sc = (p.scale && okc ? p.scale[gn + c] : 1.f) * acc_scale, bi = p.scale && okc ? p.bias[gn + c] : 0.f;
            r[c] = apply_act(fmaf(v[c], sc, bi), p.act);
            if (rs && okc) r[c] += rs[c];
        }
    }
    if (vec) {
        *reinterpret_cast<f32x4*>(o) = r;
        if (p.out_sp) uavsal_store_split4(p.out_sp + ((long long)img * p.o_is + pix) * p.ldos, gn, r);
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (gn + c < p.Cout) o[c] = r[c];
    }
}

int launch_splitk_reduce(const ConvK& k, float acc_scale, hipStream_t stream) {
    const long long items = (long long)k.M * ((k.Cout + 3) / 4);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, stream, k, acc_scale);
    return uavsal_launch_status();
}

// The first 64 KB of a conv's workspace (uavsal_conv_desc.sk_ws) are 16384 words in three regions that never overlap:
//   [0, UAVSAL_SK_TICKET_BASE)                  stream-K "published" flags, one per workgroup of the launch (G < UAVSAL_SK_STREAMK_MAX)
//   [UAVSAL_SK_TICKET_BASE, UAVSAL_SK_ERR_WORD) per-tile ticket counters of the in-launch K-share reduction (tiles 10 / 11;
//                                               tile count < UAVSAL_SK_TICKET_MAX)
//   UAVSAL_SK_ERR_WORD                          the fallback error word, when the descriptor carries none of its own
// Both users leave their region zero when a launch ends.  A stream-K wait that gave up leaves flags behind in the FIRST region
// only (the host re-zeroes the block before the next forward: Engine.check): the ticket counters of later launches of the
// same run are not touched by it.
#define UAVSAL_SK_FLAG_WORDS (65536 / 4)
#define UAVSAL_SK_ERR_WORD (UAVSAL_SK_FLAG_WORDS - 1)
#define UAVSAL_SK_TICKET_BASE 4096
#define UAVSAL_SK_STREAMK_MAX UAVSAL_SK_TICKET_BASE
#define UAVSAL_SK_TICKET_MAX (UAVSAL_SK_ERR_WORD - UAVSAL_SK_TICKET_BASE)
static_assert(UAVSAL_SK_TICKET_BASE + UAVSAL_SK_TICKET_MAX <= UAVSAL_SK_ERR_WORD, "ticket indices stay below the fallback error word");
#define UAVSAL_SPLIT_REF_SLOTS 512          /* dwproj_kernel's narrow instance: two workgroups per CU */
#define UAVSAL_SPLIT_REF_SLOTS_64 1024      /* the register-staged 64 x 64 tiles (32 KB of LDS): four per CU */

// resident workgroups per CU for one kernel instantiation (cached; queried once, outside any capture)
template <typename K>
int resident_grid(K kernel, int smem, int threads = 256) {
    int per_cu = 0, cus = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, smem) != hipSuccess || per_cu <= 0) per_cu = 1;
    return per_cu * cus;
}

}  // namespace

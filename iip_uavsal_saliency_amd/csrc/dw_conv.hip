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

namespace {

struct DwK {
    const float* in;
    const float* w9c;
    const float* scale;
    const float* bias;
    float* out;
    int ldi, ldo, H, W, Ho, Wo, C4, dil, act;
    int tiles_x, tiles_y, n_img;
    long long total;   // work items = n_img * tiles_y * tiles_x * C4
    int nblk;
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
            *reinterpret_cast<f32x4*>(p.out + (((size_t)n * p.Ho + oy) * p.Wo + ox) * p.ldo + c) = v;
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
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy + (ky - 1) * p.dil;
        if (iy < 0 || iy >= p.H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = ox + (kx - 1) * p.dil;
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

}  // namespace

extern "C" int uavsal_dw3x3(const uavsal_dw_desc* d, uavsal_stream_t stream) {
    if (!d || !d->in || !d->w9c || !d->scale || !d->bias || !d->out) return UAVSAL_EINVAL;
    if (d->n_img <= 0 || d->H <= 0 || d->W <= 0 || d->C <= 0) return UAVSAL_EINVAL;
    if ((d->C & 3) || (d->ldi & 3) || (d->ldo & 3) || d->ldi < d->C || d->ldo < d->C) return UAVSAL_EALIGN;
    if (!uavsal_aligned16(d->in) || !uavsal_aligned16(d->out) || !uavsal_aligned16(d->w9c) ||
        !uavsal_aligned16(d->scale) || !uavsal_aligned16(d->bias)) return UAVSAL_EALIGN;
    if (d->stride != 1 && d->stride != 2) return UAVSAL_ESHAPE;
    if (d->dilation < 1 || (d->stride == 2 && d->dilation != 1)) return UAVSAL_ESHAPE;
    DwK k;
    k.in = d->in; k.w9c = d->w9c; k.scale = d->scale; k.bias = d->bias; k.out = d->out;
    k.ldi = d->ldi; k.ldo = d->ldo; k.H = d->H; k.W = d->W;
    k.Ho = (d->H - 1) / d->stride + 1; k.Wo = (d->W - 1) / d->stride + 1;
    k.C4 = d->C / 4; k.dil = d->dilation; k.act = d->act; k.n_img = d->n_img;
    k.tiles_x = k.tiles_y = 0; k.total = 0; k.nblk = 0;
    hipStream_t s = (hipStream_t)stream;
    if (d->dilation != 1) {
        k.total = (long long)k.n_img * k.Ho * k.Wo * k.C4;
        const long long nblk = (k.total + 255) / 256;
        if (nblk > 0x7fffffffLL) return UAVSAL_ESHAPE;
        k.nblk = (int)nblk;
        hipLaunchKernelGGL(dw3x3_dilated_kernel, dim3(k.nblk), dim3(256), 0, s, k);
        return uavsal_launch_status();
    }
    if (d->stride == 1) {
        // 4x4 patches need ~2x fewer loads per output, but on the 12x20 / 23x40 maps they leave most of the
        // chip without a workgroup: below 2 workgroups per CU's worth of patches use 2x2 ones (4x the threads)
        const long long wg44 = ((long long)k.n_img * ((k.Ho + 3) / 4) * ((k.Wo + 3) / 4) * k.C4 + 255) / 256;
        if (wg44 < 512) return launch_dw<1, 2, 2>(k, s);
        return launch_dw<1, 4, 4>(k, s);
    }
    return launch_dw<2, 2, 2>(k, s);
}

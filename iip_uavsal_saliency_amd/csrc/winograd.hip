// Winograd F(2x2, 3x3) and F(4x4, 3x3) transforms for the path's dense 3x3 convolutions (stride 1, padding 1: conv_last of the SRF-Net
// head, reference model.py:155-156, and the ConvTWA gate convolution, model_convlstm.py:276-292), fp32 throughout.
//
//   out(2x2 tile) = A^T [ sum_c (G g G^T) .* (B^T d B) ] A        d = 4x4 input patch, g = 3x3 filter
//
// 16 multiplications per 4 outputs instead of 36: the sixteen element-wise products over the channels are sixteen
// independent GEMMs [tiles x Cin] . [Cin x Cout] -- they run as ONE launch of the ordinary fp32 GEMM with the sixteen
// transform planes as "images" and per-image weights (uavsal_conv_desc.w_group_stride) -- so a 3x3 conv costs 2.25x
// fewer MFMA FLOPs, which is what bounds it (the fp32 matrix rate, at the clock the chip holds: profiles/r3_gemm_k32.md),
// for two memory-bound transform launches.  Numerics of F(2x2): coefficients 0, +-1, +-1/2 only; against direct fp32 convolution
// the saliency map moves by 5e-5 (oracle experiment at 360x640, 8- and 20-frame calls: profiles/r3_winograd.md).  F(4x4), which
// the engine takes by default for the all-frames convolutions and, from four clips up, for the recurrence steps, is described below.
//
//   uavsal_wino_input : NHWC activation -> V[P*P][Mp][C],  V_k[tile][c] = (B^T d B)_k        one thread = (tile, 4 channels)
//   uavsal_wino_output: M[P*P][Mp][C]   -> NHWC output,    y = A^T m A, then BN / ReLU6 / residual or the ConvTWA update
// uavsal_wino_desc.R = 2: 2x2 output tiles, P = 4, sixteen planes, 2.25x fewer multiplications than direct convolution;
// R = 4: 4x4 output tiles, P = 6, thirty-six planes, 4x fewer (interpolation points 0, +-1, +-2: coefficients up to 8, a
// single conv is ~20x less accurate than direct fp32 -- 2e-5 against 1e-6 at K = 576 -- the saliency map moves by 5.3e-5).
// tile = (image, ty, tx) in row-major order; planes are Mp rows apart (Mp % 128 == 0: every plane is whole GEMM tiles;
// rows past the tiles are never written and never read back: the GEMM multiplies whatever the scratch holds there (GEMM rows
// are independent, so stale -- even non-finite -- padding rows cannot reach a real output row) and the output transform only
// reads the rows of real tiles.  The engine nevertheless zero-fills its V scratch once at allocation.
#include "common.h"

namespace {

struct WinoK {
    const float* in; float* out;
    const float* scale; const float* bias; const float* res; const float* aux; const float* hprev;
    long long in_is, out_is, res_is, aux_is, h_is;      // image strides in pixels
    long long Mp;
    int ldi, ldo, ldr, ldx, ldh;
    int n_img, H, W, C4, ty, tx, act, epi;
    long long total;
    // input transform only: the input as channel segments of other tensors (uavsal_wino_desc.n_seg)
    int nseg;
    const float* sin[3];
    int sld[3], sq1[3], sH[3], sW[3];          // sq1: one past the segment's last channel quad
    float ssy[3], ssx[3];                       // align_corners scales (source - 1) / (H - 1), as uavsal_bilinear_ac computes them
};

// one-dimensional transforms (applied to rows, then to columns)
template <int R> struct WinoT;
template <> struct WinoT<2> {
    static constexpr int P = 4;
    __device__ static __forceinline__ void bt(const f32x4 (&d)[4], f32x4 (&t)[4]) {      // B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
        t[0] = d[0] - d[2]; t[1] = d[1] + d[2]; t[2] = d[2] - d[1]; t[3] = d[1] - d[3];
    }
    __device__ static __forceinline__ void at(const f32x4 (&m)[4], f32x4 (&y)[2]) {      // A^T = [1 1 1 0; 0 1 -1 -1]
        y[0] = m[0] + m[1] + m[2]; y[1] = m[1] - m[2] - m[3];
    }
};
template <> struct WinoT<4> {
    static constexpr int P = 6;
    __device__ static __forceinline__ void bt(const f32x4 (&d)[6], f32x4 (&t)[6]) {
        // B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
        t[0] = 4.f * d[0] - 5.f * d[2] + d[4];
        t[1] = (d[3] + d[4]) - 4.f * (d[1] + d[2]);
        t[2] = 4.f * (d[1] - d[2]) + (d[4] - d[3]);
        t[3] = 2.f * (d[3] - d[1]) + (d[4] - d[2]);
        t[4] = 2.f * (d[1] - d[3]) + (d[4] - d[2]);
        t[5] = 4.f * d[1] - 5.f * d[3] + d[5];
    }
    __device__ static __forceinline__ void at(const f32x4 (&m)[6], f32x4 (&y)[4]) {
        // A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
        const f32x4 s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
        y[0] = m[0] + s12 + s34;
        y[1] = d12 + 2.f * d34;
        y[2] = s12 + 4.f * s34;
        y[3] = d12 + 8.f * d34 + m[5];
    }
};

// SEG: the transform's input is a VIRTUAL concat of up to three tensors along the channels, each either at the map's size (read as
// is) or on a smaller map and then resized on the fly exactly as uavsal_bilinear_ac does (ATen upsample_bilinear2d, align_corners) --
// conv_last of the SRF-Net reads cat[up(x5), up(x4), lv3] (reference model.py:151-156) without the two resize launches and without
// the 448-channel concat buffer ever existing (51.6 MB written and read back per 8 frames at 360x640).
template <int R, bool SEG>
__global__ __launch_bounds__(256) void wino_input_kernel(const WinoK p) {
    constexpr int P = WinoT<R>::P;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= p.total) return;
    const int c4 = (int)(idx % p.C4);
    const long long tile = idx / p.C4;
    const int tpi = p.ty * p.tx;
    const int n = (int)(tile / tpi);
    const int r = (int)(tile - (long long)n * tpi);
    const int tyi = r / p.tx, txi = r - tyi * p.tx;
    const int y0 = R * tyi - 1, x0 = R * txi - 1;
    const float* base = p.in + (size_t)n * p.in_is * p.ldi + c4 * 4;
    int ldi = p.ldi, sW = p.W, sH = p.H;
    bool resize = false;
    float sy = 0.f, sx = 0.f;
    if (SEG) {
        const int s = c4 < p.sq1[0] ? 0 : (c4 < p.sq1[1] ? 1 : 2);
        const int q0 = s == 0 ? 0 : (s == 1 ? p.sq1[0] : p.sq1[1]);
        ldi = s == 0 ? p.sld[0] : (s == 1 ? p.sld[1] : p.sld[2]);
        sH = s == 0 ? p.sH[0] : (s == 1 ? p.sH[1] : p.sH[2]);
        sW = s == 0 ? p.sW[0] : (s == 1 ? p.sW[1] : p.sW[2]);
        sy = s == 0 ? p.ssy[0] : (s == 1 ? p.ssy[1] : p.ssy[2]);
        sx = s == 0 ? p.ssx[0] : (s == 1 ? p.ssx[1] : p.ssx[2]);
        const float* b0 = s == 0 ? p.sin[0] : (s == 1 ? p.sin[1] : p.sin[2]);
        base = b0 + (size_t)n * sH * sW * ldi + (c4 - q0) * 4;
        resize = sH != p.H || sW != p.W;
    }
    // rows of the patch on the source map (resized segments): y0 / y1 / weight, as bilinear_kernel computes them
    int ry0[P], ry1[P];
    float rly[P];
    if (SEG && resize) {
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const int y = y0 + i;
            const float fy = sy * (float)y;
            const int a = (int)fy;
            ry0[i] = a; ry1[i] = a + (a < sH - 1 ? 1 : 0); rly[i] = fy - (float)a;
        }
    }
    f32x4 t[P][P];                                       // t[i][j] = (B^T d)[i][j], built column by column
#pragma unroll
    for (int j = 0; j < P; ++j) {
        const int x = x0 + j;
        f32x4 col[P], tc[P];
        int cx0 = 0, cx1 = 0;
        float lx = 0.f;
        if (SEG && resize) {
            const float fx = sx * (float)x;
            cx0 = (int)fx; cx1 = cx0 + (cx0 < sW - 1 ? 1 : 0); lx = fx - (float)cx0;
        }
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const int y = y0 + i;
            const bool ok = y >= 0 && y < p.H && x >= 0 && x < p.W;
            if (SEG && resize) {
                col[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (ok) {
                    const float ly = rly[i], hy = 1.f - ly, hx = 1.f - lx;
                    const f32x4 v00 = *reinterpret_cast<const f32x4*>(base + ((size_t)ry0[i] * sW + cx0) * ldi);
                    const f32x4 v01 = *reinterpret_cast<const f32x4*>(base + ((size_t)ry0[i] * sW + cx1) * ldi);
                    const f32x4 v10 = *reinterpret_cast<const f32x4*>(base + ((size_t)ry1[i] * sW + cx0) * ldi);
                    const f32x4 v11 = *reinterpret_cast<const f32x4*>(base + ((size_t)ry1[i] * sW + cx1) * ldi);
                    col[i] = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
                }
            } else {
                col[i] = ok ? *reinterpret_cast<const f32x4*>(base + ((size_t)y * p.W + x) * ldi) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
        WinoT<R>::bt(col, tc);
#pragma unroll
        for (int i = 0; i < P; ++i) t[i][j] = tc[i];
    }
    float* o = p.out + (size_t)tile * p.ldo + c4 * 4;
    const size_t plane = (size_t)p.Mp * p.ldo;
#pragma unroll
    for (int i = 0; i < P; ++i) {                        // (.) B = the same transform along the row
        f32x4 v[P];
        WinoT<R>::bt(t[i], v);
#pragma unroll
        for (int j = 0; j < P; ++j) *reinterpret_cast<f32x4*>(o + (size_t)(P * i + j) * plane) = v[j];
    }
}

__device__ __forceinline__ f32x4 sigmoid4(f32x4 z) {
    f32x4 g;
    g.x = 1.f / (1.f + expf(-z.x)); g.y = 1.f / (1.f + expf(-z.y));
    g.z = 1.f / (1.f + expf(-z.z)); g.w = 1.f / (1.f + expf(-z.w));
    return g;
}

template <int R>
__global__ __launch_bounds__(256) void wino_output_kernel(const WinoK p) {
    constexpr int P = WinoT<R>::P;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= p.total) return;
    const int c4 = (int)(idx % p.C4);
    const long long tile = idx / p.C4;
    const int tpi = p.ty * p.tx;
    const int n = (int)(tile / tpi);
    const int r = (int)(tile - (long long)n * tpi);
    const int tyi = r / p.tx, txi = r - tyi * p.tx;
    const float* mi = p.in + (size_t)tile * p.ldi + c4 * 4;
    const size_t plane = (size_t)p.Mp * p.ldi;
    f32x4 s[R][P];                                       // s = A^T m, built column by column
#pragma unroll
    for (int j = 0; j < P; ++j) {
        f32x4 col[P], sc_[R];
#pragma unroll
        for (int i = 0; i < P; ++i) col[i] = *reinterpret_cast<const f32x4*>(mi + (size_t)(P * i + j) * plane);
        WinoT<R>::at(col, sc_);
#pragma unroll
        for (int a = 0; a < R; ++a) s[a][j] = sc_[a];
    }
    f32x4 y[R][R];
#pragma unroll
    for (int a = 0; a < R; ++a) WinoT<R>::at(s[a], y[a]);
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
    if (p.scale) {
        sc = *reinterpret_cast<const f32x4*>(p.scale + c4 * 4);
        bi = *reinterpret_cast<const f32x4*>(p.bias + c4 * 4);
    }
    const float lo = p.act == UAVSAL_ACT_RELU6 ? 0.f : -3.0e38f, hi = p.act == UAVSAL_ACT_RELU6 ? 6.f : 3.0e38f;
#pragma unroll
    for (int a = 0; a < R; ++a)
#pragma unroll
        for (int b = 0; b < R; ++b) {
            const int yy = R * tyi + a, xx = R * txi + b;
            if (yy >= p.H || xx >= p.W) continue;
            const size_t pix = (size_t)yy * p.W + xx;
            f32x4 v = y[a][b];
            if (p.epi == UAVSAL_EPI_TWA) {      // gate = sigmoid(conv(h) + W_x x_t); h_t = gate x_t + (1 - gate) h_{t-1}
                const f32x4 z = v + *reinterpret_cast<const f32x4*>(p.aux + ((size_t)n * p.aux_is + pix) * p.ldx + c4 * 4);
                const f32x4 xt = *reinterpret_cast<const f32x4*>(p.res + ((size_t)n * p.res_is + pix) * p.ldr + c4 * 4);
                const f32x4 hp = *reinterpret_cast<const f32x4*>(p.hprev + ((size_t)n * p.h_is + pix) * p.ldh + c4 * 4);
                const f32x4 g = sigmoid4(z);
                v = g * xt + (1.f - g) * hp;
            } else {
                v.x = __builtin_amdgcn_fmed3f(fmaf(v.x, sc.x, bi.x), lo, hi); v.y = __builtin_amdgcn_fmed3f(fmaf(v.y, sc.y, bi.y), lo, hi);
                v.z = __builtin_amdgcn_fmed3f(fmaf(v.z, sc.z, bi.z), lo, hi); v.w = __builtin_amdgcn_fmed3f(fmaf(v.w, sc.w, bi.w), lo, hi);
                if (p.res) v += *reinterpret_cast<const f32x4*>(p.res + ((size_t)n * p.res_is + pix) * p.ldr + c4 * 4);
            }
            *reinterpret_cast<f32x4*>(p.out + ((size_t)n * p.out_is + pix) * p.ldo + c4 * 4) = v;
        }
}

int fill(const uavsal_wino_desc* d, WinoK& k, bool input) {
    if (!d || !d->out) return UAVSAL_EINVAL;
    const bool seg = input && d->n_seg > 0;
    if (!seg && !d->in) return UAVSAL_EINVAL;
    if (d->n_seg < 0 || d->n_seg > 3 || (!input && d->n_seg)) return UAVSAL_EINVAL;
    if (d->n_img <= 0 || d->H <= 0 || d->W <= 0 || d->C <= 0) return UAVSAL_EINVAL;
    if ((d->C & 3) || (d->ldo & 3) || d->ldo < d->C) return UAVSAL_EALIGN;
    if (!seg && ((d->ldi & 3) || d->ldi < d->C || !uavsal_aligned16(d->in))) return UAVSAL_EALIGN;
    if (!uavsal_aligned16(d->out)) return UAVSAL_EALIGN;
    k.nseg = 0;
    for (int s = 0; s < 3; ++s) { k.sin[s] = nullptr; k.sld[s] = 0; k.sq1[s] = 0x7fffffff; k.sH[s] = d->H; k.sW[s] = d->W; k.ssy[s] = k.ssx[s] = 0.f; }
    if (seg) {
        int c = 0;
        for (int s = 0; s < d->n_seg; ++s) {
            if (!d->seg_in[s] || d->seg_c[s] <= 0 || d->seg_H[s] <= 0 || d->seg_W[s] <= 0) return UAVSAL_EINVAL;
            if ((d->seg_c[s] & 3) || (d->seg_ld[s] & 3) || d->seg_ld[s] < d->seg_c[s] || !uavsal_aligned16(d->seg_in[s])) return UAVSAL_EALIGN;
            c += d->seg_c[s];
            k.sin[s] = d->seg_in[s]; k.sld[s] = d->seg_ld[s]; k.sq1[s] = c / 4; k.sH[s] = d->seg_H[s]; k.sW[s] = d->seg_W[s];
            k.ssy[s] = d->H > 1 ? (float)(d->seg_H[s] - 1) / (float)(d->H - 1) : 0.f;
            k.ssx[s] = d->W > 1 ? (float)(d->seg_W[s] - 1) / (float)(d->W - 1) : 0.f;
        }
        if (c != d->C) return UAVSAL_ESHAPE;
        k.sq1[d->n_seg - 1] = 0x7fffffff;
        k.nseg = d->n_seg;
    }
    if (d->R != 2 && d->R != 4) return UAVSAL_ESHAPE;
    k.ty = (d->H + d->R - 1) / d->R; k.tx = (d->W + d->R - 1) / d->R;
    const long long tiles = (long long)d->n_img * k.ty * k.tx;
    if ((d->Mp & 127) || d->Mp < tiles) return UAVSAL_ESHAPE;
    k.in = d->in; k.out = d->out; k.ldi = d->ldi; k.ldo = d->ldo; k.Mp = d->Mp;
    k.n_img = d->n_img; k.H = d->H; k.W = d->W; k.C4 = d->C / 4;
    const long long hw = (long long)d->H * d->W;
    k.in_is = d->in_img_stride > 0 ? d->in_img_stride : hw;
    k.out_is = d->out_img_stride > 0 ? d->out_img_stride : hw;
    k.scale = k.bias = k.res = k.aux = k.hprev = nullptr;
    k.res_is = k.aux_is = k.h_is = hw; k.ldr = k.ldx = k.ldh = 0; k.act = UAVSAL_ACT_NONE; k.epi = UAVSAL_EPI_AFFINE;
    if (!input) {
        if ((d->scale == nullptr) != (d->bias == nullptr)) return UAVSAL_EINVAL;
        if (d->act != UAVSAL_ACT_NONE && d->act != UAVSAL_ACT_RELU6) return UAVSAL_ESHAPE;
        if (d->epi != UAVSAL_EPI_AFFINE && d->epi != UAVSAL_EPI_TWA) return UAVSAL_ESHAPE;
        k.scale = d->scale; k.bias = d->bias; k.act = d->act; k.epi = d->epi;
        if (d->res) {
            if ((d->ldr & 3) || d->ldr < d->C || !uavsal_aligned16(d->res)) return UAVSAL_EALIGN;
            k.res = d->res; k.ldr = d->ldr; k.res_is = d->res_img_stride > 0 ? d->res_img_stride : hw;
        }
        if (d->epi == UAVSAL_EPI_TWA) {
            if (!d->res || !d->aux || !d->hprev) return UAVSAL_EINVAL;
            if ((d->ldx & 3) || (d->ldh & 3) || d->ldx < d->C || d->ldh < d->C || !uavsal_aligned16(d->aux) || !uavsal_aligned16(d->hprev))
                return UAVSAL_EALIGN;
            k.aux = d->aux; k.ldx = d->ldx; k.aux_is = d->aux_img_stride > 0 ? d->aux_img_stride : hw;
            k.hprev = d->hprev; k.ldh = d->ldh; k.h_is = d->h_img_stride > 0 ? d->h_img_stride : hw;
        }
        if (d->scale && (!uavsal_aligned16(d->scale) || !uavsal_aligned16(d->bias))) return UAVSAL_EALIGN;
    }
    k.total = tiles * k.C4;
    if ((k.total + 255) / 256 > 0x7fffffffLL) return UAVSAL_ESHAPE;
    return 0;
}

}  // namespace

extern "C" int uavsal_wino_input(const uavsal_wino_desc* d, uavsal_stream_t stream) {
    WinoK k;
    const int e = fill(d, k, true);
    if (e) return e;
    const dim3 grid((unsigned)((k.total + 255) / 256));
    if (k.nseg) {
        if (d->R == 4) hipLaunchKernelGGL((wino_input_kernel<4, true>), grid, dim3(256), 0, (hipStream_t)stream, k);
        else hipLaunchKernelGGL((wino_input_kernel<2, true>), grid, dim3(256), 0, (hipStream_t)stream, k);
    } else if (d->R == 4) hipLaunchKernelGGL((wino_input_kernel<4, false>), grid, dim3(256), 0, (hipStream_t)stream, k);
    else hipLaunchKernelGGL((wino_input_kernel<2, false>), grid, dim3(256), 0, (hipStream_t)stream, k);
    return uavsal_launch_status();
}

extern "C" int uavsal_wino_output(const uavsal_wino_desc* d, uavsal_stream_t stream) {
    WinoK k;
    const int e = fill(d, k, false);
    if (e) return e;
    if (d->R == 4) hipLaunchKernelGGL(wino_output_kernel<4>, dim3((unsigned)((k.total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, k);
    else hipLaunchKernelGGL(wino_output_kernel<2>, dim3((unsigned)((k.total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, k);
    return uavsal_launch_status();
}

// Output post-processing of the reference caller, on device:
//   postprocess_predictions (reference utils_data.py:289-303): bilinear resize of each h x w
//   saliency map to the source frame size with cv2.resize's INTER_LINEAR rule (half-pixel centres:
//   src = (dst + 0.5) * src/dst - 0.5, edge-replicated), centre crop of the longer side,
//   `img / max(img) * 255`; then np2mat / im2uint8 (utils_data.py:68-82): clip to [0,255] and
//   round half to even -> uint8.   (Demo_Test.py:89-91)
// Two launches: per-frame max of the resized+cropped map (block reduce + atomic max on the
// float bits, all values are > 0), then normalise + quantise.
#include "common.h"

namespace {

struct PostK {
    const float* in; unsigned char* out; unsigned int* maxbits;
    int h, w, H, W, Hr, Wr, y0, x0;     // resized size (Hr, Wr) and crop origin
    double sy, sx;                      // src/dst scale per axis (cv2 computes these in double)
    long long per_img;
};

__device__ __forceinline__ float sample(const PostK& p, const float* img, int oy, int ox) {
    const int ry = oy + p.y0, rx = ox + p.x0;
    float fy = (float)((ry + 0.5) * p.sy - 0.5);
    float fx = (float)((rx + 0.5) * p.sx - 0.5);
    int sy = (int)floorf(fy), sx = (int)floorf(fx);
    fy -= sy; fx -= sx;
    if (sy < 0) { sy = 0; fy = 0.f; }
    if (sy >= p.h - 1) { sy = p.h - 1; fy = 0.f; }
    if (sx < 0) { sx = 0; fx = 0.f; }
    if (sx >= p.w - 1) { sx = p.w - 1; fx = 0.f; }
    const int sy1 = sy + (sy < p.h - 1 ? 1 : 0), sx1 = sx + (sx < p.w - 1 ? 1 : 0);
    const float a0 = 1.f - fx, a1 = fx, b0 = 1.f - fy, b1 = fy;
    // cv2: horizontal pass on the two source rows, then the vertical blend
    const float t0 = __fadd_rn(__fmul_rn(img[sy * p.w + sx], a0), __fmul_rn(img[sy * p.w + sx1], a1));
    const float t1 = __fadd_rn(__fmul_rn(img[sy1 * p.w + sx], a0), __fmul_rn(img[sy1 * p.w + sx1], a1));
    return __fadd_rn(__fmul_rn(t0, b0), __fmul_rn(t1, b1));
}

__global__ __launch_bounds__(256) void post_max_kernel(const PostK p) {
    const int n = blockIdx.y;
    const float* img = p.in + (size_t)n * p.h * p.w;
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < p.per_img; i += (long long)gridDim.x * 256) {
        const int oy = (int)(i / p.W), ox = (int)(i - (long long)oy * p.W);
        m = fmaxf(m, sample(p, img, oy, ox));
    }
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    __shared__ float wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
        atomicMax(p.maxbits + n, __float_as_uint(m));      // positive floats order like their bits
    }
}

__global__ __launch_bounds__(256) void post_quant_kernel(const PostK p) {
    const int n = blockIdx.y;
    const float* img = p.in + (size_t)n * p.h * p.w;
    const float mx = __uint_as_float(p.maxbits[n]);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < p.per_img; i += (long long)gridDim.x * 256) {
        const int oy = (int)(i / p.W), ox = (int)(i - (long long)oy * p.W);
        float v = __fmul_rn(__fdiv_rn(sample(p, img, oy, ox), mx), 255.f);
        v = fminf(fmaxf(v, 0.f), 255.f);
        p.out[(size_t)n * p.per_img + i] = (unsigned char)rintf(v);
    }
}

}  // namespace

extern "C" int uavsal_postprocess(const uavsal_post_desc* d, uavsal_stream_t stream) {
    if (!d || !d->in || !d->out || !d->scratch) return UAVSAL_EINVAL;
    if (d->n_img <= 0 || d->h <= 0 || d->w <= 0 || d->H <= 0 || d->W <= 0) return UAVSAL_EINVAL;
    PostK k;
    k.in = d->in; k.out = d->out; k.maxbits = (unsigned int*)d->scratch;
    k.h = d->h; k.w = d->w; k.H = d->H; k.W = d->W;
    const double rows_rate = (double)d->H / d->h, cols_rate = (double)d->W / d->w;
    if (rows_rate > cols_rate) {             // utils_data.py:294-297
        k.Hr = d->H; k.Wr = (int)(((long long)d->w * d->H) / d->h);
        k.y0 = 0; k.x0 = (k.Wr - d->W) / 2;
    } else {                                 // utils_data.py:298-301
        k.Wr = d->W; k.Hr = (int)(((long long)d->h * d->W) / d->w);
        k.x0 = 0; k.y0 = (k.Hr - d->H) / 2;
    }
    if (k.Hr < d->H || k.Wr < d->W) return UAVSAL_ESHAPE;
    k.sy = (double)d->h / k.Hr; k.sx = (double)d->w / k.Wr;
    k.per_img = (long long)d->H * d->W;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(d->scratch, 0, sizeof(unsigned int) * d->n_img, s);
    if (e != hipSuccess) return (int)e;
    long long bx = (k.per_img + 255) / 256;
    if (bx > 512) bx = 512;
    dim3 grid((unsigned)bx, (unsigned)d->n_img);
    hipLaunchKernelGGL(post_max_kernel, grid, dim3(256), 0, s, k);
    hipLaunchKernelGGL(post_quant_kernel, grid, dim3(256), 0, s, k);
    return uavsal_launch_status();
}

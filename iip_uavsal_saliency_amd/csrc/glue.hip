// Small memory-bound kernels around the convolutions: the 3->32 stride-2 stem reading the
// caller's NCHW frames, align_corners bilinear resize into channel slices, the temporal
// neighbour differences of teConv_sub, the per-chunk time sum of the context prior and the
// NCHW <-> NHWC layout changes at the call boundary.  Reference lines: see uavsal_hip.h.
#include "common.h"

namespace {

// ---------------------------------------------------------------- stem 3x3 s2, 3 -> 32
struct StemK {
    const float* in; const uint8_t* in_u8; const float* w; const float* scale; const float* bias;
    float* out; int ldo, H, W, Ho, Wo, tiles_x, tiles_y;
    float mean[3], stdv[3];
};

// A workgroup owns ST_TY x ST_TX output pixels of one image.  The (2 ST_TY + 1) x (2 ST_TX + 1) x 3 input patch is staged in LDS by
// coalesced loads (consecutive lanes = consecutive columns of a row; uint8 frames are normalised here), then a thread computes 8
// pixels x 4 channels out of LDS: a tap's 4 weights are read once for 8 pixels, the 17 input columns of a (channel, kernel row) as
// four ds_read_b128 + one b32, and eight lanes write the 128 contiguous bytes of an output pixel.  (Round 2-4 form: every thread fetched its 81 inputs with 4-byte global loads, four lanes per
// address and 32 bytes between neighbours -- 324 wave-level loads per 256 pixels against 56 now; the kernel sat at 2.7 TB/s on
// its load instructions, not on bytes or FMAs: 48 v_pk_fma_f32 per loop body.)
constexpr int ST_TY = 4, ST_TX = 64, ST_IH = 2 * ST_TY + 1, ST_IW = 2 * ST_TX + 1, ST_PITCH = 132, ST_ROWS = 3 * ST_IH;
static_assert(ST_PITCH >= ST_IW && ST_PITCH % 4 == 0 && ST_TY * ST_TX == 256, "stem tile");

__global__ __launch_bounds__(256) void stem_kernel(const StemK p) {
    __shared__ __attribute__((aligned(16))) float wsh[27 * 32];
    __shared__ __attribute__((aligned(16))) float ssh[32], bsh[32];
    __shared__ __attribute__((aligned(16))) float xsh[ST_ROWS * ST_PITCH];
    const int tid = threadIdx.x;
    int b = blockIdx.x;
    const int tx = b % p.tiles_x; b /= p.tiles_x;
    const int ty = b % p.tiles_y;
    const int n = b / p.tiles_y;
    const int oy0 = ty * ST_TY, ox0 = tx * ST_TX;
    const int iy0 = 2 * oy0 - 1, ix0 = 2 * ox0 - 1;
    const size_t plane = (size_t)p.H * p.W;
    constexpr int NST = (ST_ROWS * ST_PITCH + 255) / 256;
    float st[NST];
#pragma unroll
    for (int i = 0; i < NST; ++i) {                      // all the loads of a thread first, then its LDS writes
        const int idx = tid + 256 * i;
        const int r = idx / ST_PITCH, c = idx - r * ST_PITCH;
        const int ci = r / ST_IH, iy = iy0 + (r - ci * ST_IH), ix = ix0 + c;
        float x = 0.f;
        if (r < ST_ROWS && c < ST_IW && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) {
            const size_t at = ((size_t)n * 3 + ci) * plane + (size_t)iy * p.W + ix;
            if (p.in_u8) x = ((float)p.in_u8[at] / 255.0f - p.mean[ci]) / p.stdv[ci];
            else x = p.in[at];
        }
        st[i] = x;
    }
    for (int i = tid; i < 27 * 32; i += 256) wsh[i] = p.w[i];
    if (tid < 32) { ssh[tid] = p.scale[tid]; bsh[tid] = p.bias[tid]; }
#pragma unroll
    for (int i = 0; i < NST; ++i) {
        const int idx = tid + 256 * i;
        if (idx < ST_ROWS * ST_PITCH) xsh[idx] = st[i];
    }
    __syncthreads();
    // thread = (8 consecutive output pixels of a row, 4 channels): eight lanes write the 128 contiguous bytes of a pixel
    const int g = tid & 7, pxg = tid >> 3;
    const int rr = pxg / (ST_TX / 8), xq = pxg - rr * (ST_TX / 8);
    const int oy = oy0 + rr, oxb = ox0 + 8 * xq;
    f32x4 acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int ci = 0; ci < 3; ++ci) {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const float* row = xsh + (ci * ST_IH + 2 * rr + ky) * ST_PITCH + 16 * xq;     // input columns 2*oxb-1 .. 2*oxb+15
            float v[17];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(row + 4 * c);
                v[4 * c] = a.x; v[4 * c + 1] = a.y; v[4 * c + 2] = a.z; v[4 * c + 3] = a.w;
            }
            v[16] = row[16];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(wsh + (ci * 9 + ky * 3 + kx) * 32 + g * 4);
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[q] = w4 * v[2 * q + kx] + acc[q];
            }
        }
    }
    if (oy >= p.Ho) return;
    const f32x4 sc = *reinterpret_cast<const f32x4*>(ssh + g * 4), bi = *reinterpret_cast<const f32x4*>(bsh + g * 4);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int ox = oxb + q;
        if (ox >= p.Wo) break;
        f32x4 r = acc[q] * sc + bi;
        r.x = fminf(fmaxf(r.x, 0.f), 6.f); r.y = fminf(fmaxf(r.y, 0.f), 6.f); r.z = fminf(fmaxf(r.z, 0.f), 6.f); r.w = fminf(fmaxf(r.w, 0.f), 6.f);
        *reinterpret_cast<f32x4*>(p.out + (((size_t)n * p.Ho + oy) * p.Wo + ox) * p.ldo + g * 4) = r;
    }
}

// ---------------------------------------------------------------- bilinear, align_corners=True
struct BilK {
    const float* in; float* out;
    int ldi, ldo, Hi, Wi, Ho, Wo, C4, src_mod, src_div;
    float sy, sx; long long total;
    _Float16* out_split; int ldos;                    // optional split shadow of the output
};

__global__ __launch_bounds__(256) void bilinear_kernel(const BilK p) {
    const long long item = (long long)blockIdx.x * 256 + threadIdx.x;
    if (item >= p.total) return;
    const int c = (int)(item % p.C4) * 4;
    long long t = item / p.C4;
    const int ox = (int)(t % p.Wo); t /= p.Wo;
    const int oy = (int)(t % p.Ho);
    const int n = (int)(t / p.Ho);
    const int ns = (n % p.src_mod) / p.src_div;
    // ATen upsample_bilinear2d (align_corners): src = scale * dst, scale = (in-1)/(out-1)
    const float fy = p.sy * oy, fx = p.sx * ox;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < p.Hi - 1 ? 1 : 0), x1 = x0 + (x0 < p.Wi - 1 ? 1 : 0);
    const float ly = fy - y0, lx = fx - x0;
    const float hy = 1.f - ly, hx = 1.f - lx;
    const float* b = p.in + (size_t)ns * p.Hi * p.Wi * p.ldi + c;
    const f32x4 v00 = *reinterpret_cast<const f32x4*>(b + ((size_t)y0 * p.Wi + x0) * p.ldi);
    const f32x4 v01 = *reinterpret_cast<const f32x4*>(b + ((size_t)y0 * p.Wi + x1) * p.ldi);
    const f32x4 v10 = *reinterpret_cast<const f32x4*>(b + ((size_t)y1 * p.Wi + x0) * p.ldi);
    const f32x4 v11 = *reinterpret_cast<const f32x4*>(b + ((size_t)y1 * p.Wi + x1) * p.ldi);
    const f32x4 r = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
    const size_t opix = ((size_t)n * p.Ho + oy) * p.Wo + ox;
    *reinterpret_cast<f32x4*>(p.out + opix * p.ldo + c) = r;
    if (p.out_split) uavsal_store_split4(p.out_split + opix * p.ldos, c, r);
}

// ---------------------------------------------------------------- temporal differences
struct TdK { const float* in; float* out; int ldi, ldo, HW, C4, L; long long total; };

__global__ __launch_bounds__(256) void tdiff_kernel(const TdK p) {
    const long long item = (long long)blockIdx.x * 256 + threadIdx.x;
    if (item >= p.total) return;
    const int c = (int)(item % p.C4) * 4;
    long long t = item / p.C4;
    const int pix = (int)(t % p.HW);
    const int n = (int)(t / p.HW);
    const int tt = n % p.L;
    const size_t img = (size_t)p.HW * p.ldi;
    const float* cur = p.in + (size_t)n * img + (size_t)pix * p.ldi + c;
    const f32x4 x = *reinterpret_cast<const f32x4*>(cur);
    f32x4 dprev, dnext;
    if (tt == 0) {
        const f32x4 nx = *reinterpret_cast<const f32x4*>(cur + img);
        dprev = nx - x; dnext = x - nx;                       // model.py:194
    } else if (tt == p.L - 1) {
        const f32x4 pv = *reinterpret_cast<const f32x4*>(cur - img);
        dprev = x - pv; dnext = pv - x;                       // model.py:198
    } else {
        const f32x4 pv = *reinterpret_cast<const f32x4*>(cur - img);
        const f32x4 nx = *reinterpret_cast<const f32x4*>(cur + img);
        dprev = x - pv; dnext = x - nx;                       // model.py:196
    }
    float* o = p.out + ((size_t)n * p.HW + pix) * p.ldo + c;
    *reinterpret_cast<f32x4*>(o) = dprev;
    *reinterpret_cast<f32x4*>(o + p.C4 * 4) = dnext;
}

// ---------------------------------------------------------------- sum over T images
struct TsK { const float* in; float* out; int ldi, ldo, HW, C4, T; long long total; };

__global__ __launch_bounds__(256) void tsum_kernel(const TsK p) {
    const long long item = (long long)blockIdx.x * 256 + threadIdx.x;
    if (item >= p.total) return;
    const int c = (int)(item % p.C4) * 4;
    long long t = item / p.C4;
    const int pix = (int)(t % p.HW);
    const int b = (int)(t / p.HW);
    const size_t img = (size_t)p.HW * p.ldi;
    const float* src = p.in + (size_t)b * p.T * img + (size_t)pix * p.ldi + c;
    f32x4 s = *reinterpret_cast<const f32x4*>(src);
    for (int k = 1; k < p.T; ++k) s += *reinterpret_cast<const f32x4*>(src + (size_t)k * img);
    *reinterpret_cast<f32x4*>(p.out + ((size_t)b * p.HW + pix) * p.ldo + c) = s;
}

// ---------------------------------------------------------------- NCHW <-> NHWC
struct LayK { const float* in; float* out; int C, HW, ld, Cpad, to_nhwc; long long total; };

__global__ __launch_bounds__(256) void layout_kernel(const LayK p) {
    // 32x32 (pixel x channel) tile transposed through LDS so both sides stay coalesced
    __shared__ float tile[32][33];
    const int tiles_c = (p.Cpad + 31) / 32;
    const int tiles_p = (p.HW + 31) / 32;
    long long b = blockIdx.x;
    const int tc = (int)(b % tiles_c); b /= tiles_c;
    const int tp = (int)(b % tiles_p);
    const int n = (int)(b / tiles_p);
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;   // 32 x 8
    if (p.to_nhwc) {
        for (int j = ly; j < 32; j += 8) {        // read: channel tc*32+j, pixels contiguous
            const int c = tc * 32 + j, px = tp * 32 + lx;
            tile[j][lx] = (c < p.C && px < p.HW) ? p.in[((size_t)n * p.C + c) * p.HW + px] : 0.f;
        }
        __syncthreads();
        for (int j = ly; j < 32; j += 8) {        // write: pixel tp*32+j, channels contiguous
            const int px = tp * 32 + j, c = tc * 32 + lx;
            if (px < p.HW && c < p.Cpad) p.out[((size_t)n * p.HW + px) * p.ld + c] = tile[lx][j];
        }
    } else {
        for (int j = ly; j < 32; j += 8) {
            const int px = tp * 32 + j, c = tc * 32 + lx;
            tile[j][lx] = (px < p.HW && c < p.C) ? p.in[((size_t)n * p.HW + px) * p.ld + c] : 0.f;
        }
        __syncthreads();
        for (int j = ly; j < 32; j += 8) {
            const int c = tc * 32 + j, px = tp * 32 + lx;
            if (c < p.C && px < p.HW) p.out[((size_t)n * p.C + c) * p.HW + px] = tile[lx][j];
        }
    }
}

inline int grid_for(long long total, int* nblk) {
    const long long b = (total + 255) / 256;
    if (b <= 0 || b > 0x7fffffffLL) return UAVSAL_ESHAPE;
    *nblk = (int)b;
    return 0;
}

}  // namespace

extern "C" int uavsal_stem_conv(const uavsal_stem_desc* d, uavsal_stream_t stream) {
    if (!d || (!d->in && !d->in_u8) || !d->w || !d->scale || !d->bias || !d->out) return UAVSAL_EINVAL;
    if (d->n_img <= 0 || d->H <= 0 || d->W <= 0) return UAVSAL_EINVAL;
    if ((d->ldo & 3) || d->ldo < 32 || !uavsal_aligned16(d->out)) return UAVSAL_EALIGN;
    StemK k;
    k.in = d->in; k.in_u8 = d->in_u8; k.w = d->w; k.scale = d->scale; k.bias = d->bias; k.out = d->out;
    k.ldo = d->ldo; k.H = d->H; k.W = d->W; k.Ho = (d->H - 1) / 2 + 1; k.Wo = (d->W - 1) / 2 + 1;
    for (int i = 0; i < 3; ++i) { k.mean[i] = d->mean[i]; k.stdv[i] = d->stdv[i]; }
    k.tiles_x = (k.Wo + ST_TX - 1) / ST_TX; k.tiles_y = (k.Ho + ST_TY - 1) / ST_TY;
    const long long nblk = (long long)d->n_img * k.tiles_y * k.tiles_x;
    if (nblk > 0x7fffffffLL) return UAVSAL_ESHAPE;
    hipLaunchKernelGGL(stem_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, k);
    return uavsal_launch_status();
}

extern "C" int uavsal_bilinear_ac(const uavsal_bilinear_desc* d, uavsal_stream_t stream) {
    if (!d || !d->in || !d->out) return UAVSAL_EINVAL;
    if (d->n_out <= 0 || d->C <= 0 || d->Hi <= 0 || d->Wi <= 0 || d->Ho <= 0 || d->Wo <= 0) return UAVSAL_EINVAL;
    if (d->src_mod <= 0 || d->src_div <= 0) return UAVSAL_EINVAL;
    if ((d->C & 3) || (d->ldi & 3) || (d->ldo & 3) || d->ldi < d->C || d->ldo < d->C) return UAVSAL_EALIGN;
    if (!uavsal_aligned16(d->in) || !uavsal_aligned16(d->out)) return UAVSAL_EALIGN;
    BilK k;
    k.in = d->in; k.out = d->out; k.ldi = d->ldi; k.ldo = d->ldo;
    k.Hi = d->Hi; k.Wi = d->Wi; k.Ho = d->Ho; k.Wo = d->Wo; k.C4 = d->C / 4;
    k.src_mod = d->src_mod; k.src_div = d->src_div;
    k.sy = d->Ho > 1 ? (float)(d->Hi - 1) / (float)(d->Ho - 1) : 0.f;
    k.sx = d->Wo > 1 ? (float)(d->Wi - 1) / (float)(d->Wo - 1) : 0.f;
    if (d->out_split && ((d->ldos & 63) || d->ldos < 2 * d->C || (d->C & 31) || ((uintptr_t)d->out_split & 127)))
        return UAVSAL_EALIGN;
    k.out_split = (_Float16*)d->out_split; k.ldos = d->ldos;
    k.total = (long long)d->n_out * d->Ho * d->Wo * k.C4;
    int nblk; int e = grid_for(k.total, &nblk); if (e) return e;
    hipLaunchKernelGGL(bilinear_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, k);
    return uavsal_launch_status();
}

extern "C" int uavsal_tdiff(const uavsal_tdiff_desc* d, uavsal_stream_t stream) {
    if (!d || !d->in || !d->out) return UAVSAL_EINVAL;
    if (d->n_img <= 0 || d->HW <= 0 || d->C <= 0) return UAVSAL_EINVAL;
    if (d->seq_len < 2 || (d->n_img % d->seq_len)) return UAVSAL_ESHAPE;   // reference raises on 1 frame
    if ((d->C & 3) || (d->ldi & 3) || (d->ldo & 3) || d->ldi < d->C || d->ldo < 2 * d->C) return UAVSAL_EALIGN;
    if (!uavsal_aligned16(d->in) || !uavsal_aligned16(d->out)) return UAVSAL_EALIGN;
    TdK k; k.in = d->in; k.out = d->out; k.ldi = d->ldi; k.ldo = d->ldo; k.HW = d->HW; k.C4 = d->C / 4; k.L = d->seq_len;
    k.total = (long long)d->n_img * d->HW * k.C4;
    int nblk; int e = grid_for(k.total, &nblk); if (e) return e;
    hipLaunchKernelGGL(tdiff_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, k);
    return uavsal_launch_status();
}

extern "C" int uavsal_tsum(const uavsal_tsum_desc* d, uavsal_stream_t stream) {
    if (!d || !d->in || !d->out) return UAVSAL_EINVAL;
    if (d->n_groups <= 0 || d->T <= 0 || d->HW <= 0 || d->C <= 0) return UAVSAL_EINVAL;
    if ((d->C & 3) || (d->ldi & 3) || (d->ldo & 3) || d->ldi < d->C || d->ldo < d->C) return UAVSAL_EALIGN;
    if (!uavsal_aligned16(d->in) || !uavsal_aligned16(d->out)) return UAVSAL_EALIGN;
    TsK k; k.in = d->in; k.out = d->out; k.ldi = d->ldi; k.ldo = d->ldo; k.HW = d->HW; k.C4 = d->C / 4; k.T = d->T;
    k.total = (long long)d->n_groups * d->HW * k.C4;
    int nblk; int e = grid_for(k.total, &nblk); if (e) return e;
    hipLaunchKernelGGL(tsum_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, k);
    return uavsal_launch_status();
}

extern "C" int uavsal_layout(const uavsal_layout_desc* d, uavsal_stream_t stream) {
    if (!d || !d->in || !d->out) return UAVSAL_EINVAL;
    if (d->n_img <= 0 || d->C <= 0 || d->HW <= 0) return UAVSAL_EINVAL;
    const int cpad = d->to_nhwc ? (d->Cpad > d->C ? d->Cpad : d->C) : d->C;
    if (d->ld < cpad) return UAVSAL_ESHAPE;
    LayK k; k.in = d->in; k.out = d->out; k.C = d->C; k.HW = d->HW; k.ld = d->ld; k.Cpad = cpad; k.to_nhwc = d->to_nhwc;
    const long long nb = (long long)d->n_img * ((d->HW + 31) / 32) * ((cpad + 31) / 32);
    if (nb > 0x7fffffffLL) return UAVSAL_ESHAPE;
    k.total = nb;
    hipLaunchKernelGGL(layout_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, k);
    return uavsal_launch_status();
}

// ---------------------------------------------------------------- error guard (see uavsal_hip.h)
namespace {
__global__ __launch_bounds__(256) void guard_kernel(const uavsal_guard_desc d) {
    const int e = *d.err;
    if (blockIdx.x == 0 && threadIdx.x == 0 && d.host_err)
        __hip_atomic_store(d.host_err, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (e == 0) return;
    const float qnan = __builtin_nanf("");
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        if (!d.buf[b]) continue;
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < d.n[b]; i += (long long)gridDim.x * 256)
            d.buf[b][i] = qnan;
    }
}
}  // namespace

extern "C" int uavsal_guard(const uavsal_guard_desc* d, uavsal_stream_t stream) {
    if (!d || !d->err) return UAVSAL_EINVAL;
    for (int b = 0; b < 3; ++b)
        if ((d->buf[b] == nullptr) != (d->n[b] == 0) || d->n[b] < 0) return UAVSAL_EINVAL;
    hipLaunchKernelGGL(guard_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, *d);
    return uavsal_launch_status();
}

// ---------------------------------------------------------------- word fill (see uavsal_hip.h)
namespace {
__global__ __launch_bounds__(256) void fill_kernel(const uavsal_fill_desc d) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < d.n; i += (long long)gridDim.x * 256) d.out[i] = d.bits;
}
}  // namespace

extern "C" int uavsal_fill(const uavsal_fill_desc* d, uavsal_stream_t stream) {
    if (!d || !d->out || d->n <= 0) return UAVSAL_EINVAL;
    if ((uintptr_t)d->out & 3) return UAVSAL_EALIGN;
    long long nblk = (d->n + 255) / 256;
    if (nblk > 4096) nblk = 4096;
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, *d);
    return uavsal_launch_status();
}

// ---------------------------------------------------------------- strided row copy (see uavsal_hip.h)
namespace {
__global__ __launch_bounds__(256) void copy_rows_kernel(const uavsal_copy_desc d) {
    const long long per_row = d.row_floats >> 2;
    const long long total = per_row * d.rows;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / per_row, c = i - r * per_row;
        *reinterpret_cast<f32x4*>(d.out + r * d.out_pitch + c * 4) =
            *reinterpret_cast<const f32x4*>(d.in + r * d.in_pitch + c * 4);
    }
}
}  // namespace

extern "C" int uavsal_copy_rows(const uavsal_copy_desc* d, uavsal_stream_t stream) {
    if (!d || !d->in || !d->out || d->rows <= 0 || d->row_floats <= 0) return UAVSAL_EINVAL;
    if ((d->row_floats & 3) || (d->in_pitch & 3) || (d->out_pitch & 3) || !uavsal_aligned16(d->in) || !uavsal_aligned16(d->out))
        return UAVSAL_EALIGN;
    if (d->in_pitch < d->row_floats || d->out_pitch < d->row_floats) return UAVSAL_ESHAPE;
    const long long total = (d->row_floats >> 2) * d->rows;
    long long nblk = (total + 255) / 256;
    if (nblk > 2048) nblk = 2048;
    hipLaunchKernelGGL(copy_rows_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, *d);
    return uavsal_launch_status();
}

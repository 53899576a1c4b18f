/*
 * uavsal_hip.h -- C ABI of libuavsal_hip.so, the MI355X (gfx950) kernels behind
 * UAVSal.forward.
 *
 * The reference (zhangkao/IIP_UAVSal_Saliency) is pure Python: it has no native
 * boundary of its own.  Every arithmetic step of `UAVSal.forward`
 * (model.py:341-375) is a stock torch.nn operator.  This header therefore
 * defines the native boundary the drop-in needs: one entry point per operator
 * class of the hot path, each citing the reference expression (file:line,
 * relative to the reference root) it replaces.  Plain pointers and sizes only:
 * no torch types, no allocation, no ownership transfer, no global state.
 *
 * Conventions
 *   - all tensors are fp32 device pointers, activations NHWC
 *     ([image][y][x][channel]); `ld*` is the channel stride of the buffer in
 *     floats, so a pointer + ld pair can address a channel slice of a wider
 *     buffer (this is how torch.cat, model.py:146,155,363,365, disappears);
 *   - `*_img_stride` is the distance between consecutive images in PIXELS
 *     (H*W for a dense batch; T*H*W when one time step of C clips is addressed,
 *     model_convlstm.py:368-371);
 *   - channel counts, ld's and slice offsets are multiples of 4 floats (16 B);
 *   - `stream` is a hipStream_t passed as void*; kernels are launched on it and
 *     nothing synchronises;
 *   - every function returns 0 on success, a positive hipError_t if the launch
 *     failed, or a negative UAVSAL_E* code if the arguments were rejected.
 *
 * Built for gfx950 only.  No CUDA path, no CPU fallback.
 */
#ifndef UAVSAL_HIP_H
#define UAVSAL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UAVSAL_ABI_VERSION 20

/* argument errors */
#define UAVSAL_EINVAL   (-1)  /* null pointer / non-positive size */
#define UAVSAL_EALIGN   (-2)  /* channel count / ld / pointer not 16-byte aligned */
#define UAVSAL_ESHAPE   (-3)  /* shape not supported by the kernel (see entry point) */
#define UAVSAL_ESTATE   (-4)  /* plan used in the wrong state */
#define UAVSAL_EDEVICE  (-5)  /* a kernel reported that it could not complete its work (uavsal_conv_desc.err):
                                 the outputs of that run are invalid and were overwritten with NaN */

/* bits of the device error word (uavsal_conv_desc.err) */
#define UAVSAL_ERR_STREAMK  1 /* a stream-K owner gave up waiting for a partial tile another workgroup owed it */

/* GEMM operand precision (how fp32 activations/weights are fed to the matrix cores) */
#define UAVSAL_PREC_F32     0 /* v_mfma_f32_32x32x2_f32: exact fp32 fma chain */
#define UAVSAL_PREC_BF16X3  1 /* 3x v_mfma_f32_32x32x16_bf16 on a hi/lo bf16 split, fp32 accumulate */
#define UAVSAL_PREC_BF16    2 /* 1x bf16 MFMA, fp32 accumulate */
#define UAVSAL_PREC_F16X3   3 /* 3x v_mfma_f32_32x32x16_f16 on a hi/lo fp16 split (22 mantissa bits), fp32 accumulate;
                                 activations are pre-scaled by 2^4 (finite for any finite x: saturates near |x| = 8188), weights by 2^6 */

#define UAVSAL_ACT_NONE     0
#define UAVSAL_ACT_RELU6    1 /* nn.ReLU6, model.py:71 */
#define UAVSAL_ACT_SIGMOID  2 /* torch.sigmoid, model.py:373 */

#define UAVSAL_EPI_AFFINE   0 /* y = act(acc*scale + bias) [+ res] */
#define UAVSAL_EPI_TWA      1 /* ConvTWA gate + state update, see uavsal_conv_desc */
#define UAVSAL_EPI_LSTM     2 /* ConvLSTM gates + cell/hidden update, see uavsal_conv_desc */

typedef void* uavsal_stream_t;

/*
 * Dense 1x1 or 3x3 (stride 1, pad 1) convolution as an implicit GEMM on the
 * matrix cores, with the folded BatchNorm, activation, residual add and the
 * ConvTWA state update fused into the epilogue.
 *
 * Replaces: BasicConv2d k=1 / k=3 (model.py:65-72: Conv2d(bias=False) + BatchNorm2d
 * (eval: scale = gamma/sqrt(var+1e-5), bias = beta - mean*scale) + ReLU6); the
 * pw-linear Conv2d+BatchNorm2d tail of dwBlock (model.py:94-95) with its residual
 * (model.py:100-101); torchvision InvertedResidual's pointwise convs; STBlock's
 * `x_sp + x_te` and `x + out` adds (model.py:241,247) as post-activation residuals;
 * the final torch.sigmoid (model.py:373); and, with epi=UAVSAL_EPI_TWA,
 * ConvTWACell.forward (model_convlstm.py:276-292).
 *
 * EPI_AFFINE: out[m, n] = act(acc[m, n] * scale[n] + bias[n]) + (res ? res[m, n] : 0)
 *             (scale == NULL -> raw accumulator, bias ignored)
 * EPI_TWA:    a = h_{t-1} (Cin == Cout), w = the W[:, Cin:, :, :] half of rnn_conv,
 *             aux = conv3x3(W[:, :Cin], x_t) precomputed for this step, res = x_t:
 *             i = sigmoid(acc + aux);  out = i * res + (1 - i) * a
 * EPI_LSTM:   ConvLSTMCell.forward (model_convlstm.py:111-126, bias=False) with the x half of the
 *             conv hoisted: a = h_{t-1} (Cin = hid), Cout = 4*hid with the rows of W[:, hid:] (and
 *             of the hoisted aux) interleaved as n = 4*c + gate, gate order (i, f, o, g);
 *             res = c_{t-1} [hid], out = h_t [hid], out2 = c_t [hid]:
 *             z = acc + aux;  c_t = sigmoid(z_f)*c_{t-1} + sigmoid(z_i)*tanh(z_g);
 *             h_t = sigmoid(z_o)*tanh(c_t).   Runs on the 64x64 tile; needs 16-byte aligned ld's.
 *
 * Weights `w` are pre-packed by the host (iip_uavsal_saliency_amd/packing.py):
 *   k index = ci for taps == 1; for taps == 9 (tap = ky*3 + kx) channel-block-major, tap-minor:
 *   k = ((ci / KT) * 9 + tap) * KT + ci % KT, so that the nine taps of one channel chunk are
 *   nine consecutive K steps; rows padded to Npad = roundup(Cout, 32), K padded to
 *   Kpad = roundup(taps * Cin, KT) with zeros, KT = 16 (F32) or 32 (BF16*, F16X3);
 *   F32:    float  [Npad][Kpad]
 *   The 16-bit layouts are K-step-major, so that the weight rows one K step needs are ONE
 *   contiguous run of whole cache lines (measured +4..7 % on the split GEMMs; the LDS-DMA
 *   fp32 kernel measured 2..6 % slower with it and keeps the row-major layout):
 *   BF16:   uint16 [Kpad/32][Npad][32]     (bf16 bits, round-to-nearest-even)
 *   BF16X3: uint16 [Kpad/32][2][Npad][32]  ([.][0] = hi, [.][1] = lo = bf16(w - hi))
 *   F16X3:  as BF16X3 with fp16 bits of ws = 64*w: hi = fp16(ws), lo = fp16(ws - hi)
 *   in the BF16* / F16X3 layouts each group of 32 k's is stored in the order
 *   {0-3,16-19, 4-7,20-23, 8-11,24-27, 12-15,28-31} (matches the A staging).
 * taps == 9 requires Cin % 32 == 0.
 */
typedef struct uavsal_conv_desc {
    const float* a;      int32_t lda;  int64_t a_img_stride;
    const void*  w;
    const float* scale;  const float* bias;          /* [>= Cout] each, or NULL */
    float*       out;    int32_t ldc;  int64_t o_img_stride;
    const float* res;    int32_t ldr;  int64_t r_img_stride;   /* NULL = none */
    const float* aux;    int32_t ldx;  int64_t x_img_stride;   /* EPI_TWA only */
    int32_t n_img, H, W;         /* output == input spatial size */
    int32_t Cin, Cout, taps;     /* taps: 1 or 9 */
    int32_t prec, act, epi;
    int32_t tile;                /* 0 = auto; else 1: 128x128, 2: 128x64, 3: 128x32, 4: 64x64, 5: 128x256, 6: 256x256 (split
                                  * 16-bit precisions; others run them as 1), 7: 256x128 on 8 waves (F32; others as 1);
                                  * F32 with Cin % 32 == 0, K stages of 32 floats = one cache line per tile row
                                  * (conv_gemm_k32.hip; 3x3 weights then in the 'f32k32' K order, see `w`): 8: 128x128,
                                  * 9: 256x128 on 8 waves, 10: 128x128 as one continuous stage stream (vector affine
                                  * epilogue only), 11: 64x64.  With `sk_ws`, 8 / 10 / 11 split K over workgroups when
                                  * the tiles alone would leave most of the chip idle (8: shares + reduce launch;
                                  * 10, 11: the last share to arrive adds them in share order inside the launch).
                                  * Shapes an instance cannot take fall back: 10 -> 8 -> 1, 9 -> 7, 11 -> 4. */
    float*       out2;   int32_t ld2;                 /* EPI_LSTM only: c_t (image stride = o_img_stride) */
    /* Fused depthwise producer (taps == 1, EPI_AFFINE): when dw_w9c != NULL, `a` is the EXPANDED tensor E
     * [n_img, dw_Hin, dw_Win, Cin] of an inverted-residual block and the GEMM's A operand is computed on the
     * fly as relu6(dw_scale * depthwise3x3(E; dw_w9c, stride dw_stride, pad 1) + dw_bias), i.e. the
     * groups=hidden BasicConv2d of dwBlock (model.py:92) feeding its pw-linear conv (model.py:94) without the
     * intermediate tensor ever reaching HBM.  H, W are the OUTPUT sizes ((dw_Hin-1)/dw_stride+1, ...);
     * dw_w9c is tap-major [9][Cin] as in uavsal_dw_desc; a_img_stride counts input pixels.
     * F32 / F16X3, dw_stride 1, Cin % 16 == 0 (<= 4096) run the LDS-halo kernel (a workgroup stages the 10 x 18
     * halo of an 8 x 16 pixel patch per 16 channels, computes the depthwise in LDS and feeds the MFMA loop:
     * uavsal_conv_dwproj tells; its F16X3 form takes the weights as [Cin/16][Npad][hi 16 | lo 16] fp16 lines of
     * 64 w, packing.py 'f16x3j'); everything else the register-staged loader of round 1 (correct, latency-bound).
     * With `sk_ws` set, narrow outputs (Cout <= 32) whose tiles would leave workgroup slots empty split K over 2-4
     * workgroups per tile: raw partial sums go to the partial-tile area of `sk_ws` and a second small launch sums them
     * in a fixed order and applies BN / activation / residual (deterministic; no atomics). */
    const float* dw_w9c; const float* dw_scale; const float* dw_bias;
    int32_t dw_stride, dw_Hin, dw_Win;
    /* Optional stream-K workspace (fp32, 128x128 and 64x64 tiles; 64 KB of flags, then partial tiles): `uavsal_streamk_workspace_bytes()` bytes of device
     * memory, 16-byte aligned, ZERO-filled once by the caller and then owned by launches that are ordered
     * on one stream (the kernels leave it zeroed).  With it, launches whose tile count would idle part of
     * the chip in the last round split the K loop of some tiles across workgroups; the summation order
     * stays a function of the shape only.  NULL = whole-tile scheduling.
     * Progress: a workgroup computes the piece it owes another tile BEFORE it waits for anything, and it only
     * ever waits for workgroups with a higher block id, so the launch completes whenever the hardware starts
     * workgroups in block order (it does; HIP does not promise it).  The wait is therefore bounded
     * (`sk_spin_limit` polls): an owner that gives up leaves the missing piece OUT of its sum and sets
     * UAVSAL_ERR_STREAMK in `*err` -- the caller must treat the output as invalid and re-zero the flag block
     * (uavsal_guard / uavsal_plan_status do both checks for a plan). */
    void* sk_ws; int64_t sk_ws_bytes;
    /* Pre-split operands (UAVSAL_PREC_F16X3 only).  The "split shadow" of an fp32 NHWC activation tensor x with
     * channel stride ld (ld % 32 == 0) is an fp16 buffer [pixel][ld / 32][2][32]: for every group of 32
     * channels, 32 halves hi = fp16_rtz(16*x) followed by 32 halves lo = fp16_rtz(16*x - hi) -- exactly what the
     * register-staged kernel computes from `a` on every load, in the same number of bytes as x, laid out so
     * that the 32 channels one K step needs are ONE 128-byte line per pixel (planar hi / lo planes made every
     * request half a line and the L1 <- L2 traffic 1.5x the useful bytes: profiles/r2_presplit_pmc.md).
     * Row stride `ld*s` is in halves (2 * ld for a dense shadow); a channel slice starts at a multiple of 32.
     * When a_split is given (and Cin % 32 == 0, tile 1 / 5 / 6, vector epilogue) the GEMM stages BOTH operands
     * with LDS-DMA and does no conversion work; `w` must then be packed 'f16x3i' (packing.py: per K step and
     * output channel the same [32 hi | 32 lo] line; uavsal_conv_uses_split tells which).  `a` may be NULL
     * then unless epi == UAVSAL_EPI_TWA.
     * out_split (UAVSAL_PREC_F16X3, vector epilogue only; Cout % 32 == 0): the result is additionally written as
     * a split shadow for the GEMMs that consume it -- the fp32 copy stays for every other consumer. */
    const void* a_split;  int32_t ldas;
    void* out_split;      int32_t ldos;
    int32_t* err;            /* device word OR-ed with UAVSAL_ERR_* (NULL: the last int32 of sk_ws' 64 KB flag block) */
    int32_t sk_spin_limit;   /* polls before a stream-K owner gives up on one piece; 0 = default (1 << 22, seconds) */
    int32_t sk_debug_drop;   /* TEST HOOK: 1 + index of the stream-K workgroup that withholds its "published" flag (< 0: all do); 0 = none */
    /* Per-image weights (F32, taps == 1, tiles 8 / 11; 0 = one weight matrix for all images): image g multiplies with
     * the packed matrix at w + g * w_group_stride floats, and H * W must be a multiple of 128 so that no GEMM tile
     * straddles two images.  This is how the sixteen GEMMs of a Winograd 3x3 convolution run as one launch
     * (uavsal_wino_input / uavsal_wino_output: the images are the sixteen transform planes). */
    int64_t w_group_stride;
    /* Several convs of different inputs as ONE launch (F32, taps == 1, tile 11; 0 = off): output channels
     * [g * n_group, (g + 1) * n_group) are computed from the A columns [g * a_group_off, g * a_group_off + Cin), i.e. the
     * inputs lie side by side in the rows of `a` (lda >= groups * a_group_off) and the weight rows of the groups are stacked
     * ([Cout][Kpad], the ordinary layout).  n_group % 64 == 0, Cout % n_group == 0.  The three dilated ASPP projections
     * (model.py:142-147) run this way. */
    int32_t n_group, a_group_off;
} uavsal_conv_desc;

int uavsal_conv_gemm(const uavsal_conv_desc* d, uavsal_stream_t stream);
/* block tile `uavsal_conv_gemm` will use for this descriptor (1..11, see `tile`); no launch.  Callers pack fp32
 * 3x3 weights by it: tiles 8-11 take the K order with 32-channel blocks (k = ((ci / 32) * 9 + tap) * 32 + ci % 32,
 * packing.py 'f32k32'), the others 16-channel blocks ('f32'); 1x1 weights are the same either way. */
int uavsal_conv_tile(const uavsal_conv_desc* d);
/* 1 when `uavsal_conv_gemm` will take the pre-split LDS-DMA path for this descriptor (a_split set, shape
 * eligible) and therefore expects `w` in the 'f16x3i' packing, else 0; no launch */
int uavsal_conv_uses_split(const uavsal_conv_desc* d);
/* 0, or the LDS-halo depthwise -> projection instance this descriptor launches (no launch): its output-channel
 * tile 256 / 128 / 64 / 32 = dwproj_kernel<PREC,2,4,2,2> / <PREC,2,4,2,1> / <PREC,4,2,1,1> / <PREC,4,1,1,1> */
int uavsal_conv_dwproj(const uavsal_conv_desc* d);
/* size of the optional stream-K workspace (see uavsal_conv_desc.sk_ws) */
long long uavsal_streamk_workspace_bytes(void);
/* workgroups of the stream-K launch `uavsal_conv_gemm` will use for this descriptor, 0 = whole tiles; no launch */
int uavsal_conv_streamk_grid(const uavsal_conv_desc* d);

/*
 * Depthwise 3x3 convolution + folded BatchNorm + ReLU6, NHWC, stride 1 or 2,
 * dilation d (padding = d), one launch for the whole batch.
 * Replaces: the `groups=hidden_dim` BasicConv2d of dwBlock (model.py:92) and
 * torchvision InvertedResidual's depthwise ConvBNReLU.
 * w9c is tap-major: w9c[(ky*3+kx)*C + c] = weight[c, 0, ky, kx].
 * Ho = (H - 1) / stride + 1, Wo likewise (pad = dilation, kernel 3).
 * stride 2 requires dilation 1.
 */
typedef struct uavsal_dw_desc {
    const float* in;   int32_t ldi;
    const float* w9c;  const float* scale;  const float* bias;
    float*       out;  int32_t ldo;
    int32_t n_img, H, W, C, stride, dilation, act;
    /* optional: write the result as a split shadow (see uavsal_conv_desc) INSTEAD of fp32 -- the only consumer
     * of a depthwise output is the projection GEMM (model.py:94).  `out` may then be NULL.  dilation 1 only. */
    void* out_split;  int32_t ldos;
    /* optional: several dilated branches of the SAME map in one launch (the three dilated ASPP depthwise convs, model.py:125-127,
     * on the channel slices of their merged expand): dil_group_c > 0 (a multiple of 4, C <= 4 * dil_group_c) gives channel c the
     * dilation dil_groups[c / dil_group_c]; `dilation` is then ignored.  Stride 1, no split output. */
    int32_t dil_group_c;  int32_t dil_groups[4];
} uavsal_dw_desc;

int uavsal_dw3x3(const uavsal_dw_desc* d, uavsal_stream_t stream);
/* kernel instance `uavsal_dw3x3` will use for this descriptor; no launch.  1: dw3x3_kernel<1,4,4> (4x4 output
 * patch per thread), 2: <1,2,2>, 3: <2,2,2> (stride 2), 4: dw3x3_dilated_kernel (one pixel per thread),
 * 16 / 32 / 64: dw3x3_map_lds_kernel<CB, 256> (whole map of a CB-channel slab staged in LDS by LDS-DMA; small maps, any
 * dilation), + 512 (528 / 544 / 576) for its 512-thread instance (slabs of >= 2048 float4 items),
 * 2048: dw3x3_rowclass_kernel<256> (merged dilated branches on a map too big for 32-channel whole-map slabs: a workgroup stages
 * the rows of one residue class oy % dilation for a 64..256-channel slab) */
int uavsal_dw_variant(const uavsal_dw_desc* d);

/*
 * Depthwise 3x3 (stride 1, pad 1) + BN + ReLU6 followed by a 1x1 projection to ONE channel + BN + activation, one launch:
 *   out[pixel] = act(scale2 * sum_c w2[c] * relu6(scale[c] * dw3x3(in)[pixel, c] + bias[c]) + bias2)
 * Replaces the depthwise BasicConv2d + pw-linear conv + BN of dwBlock(256 -> 1) = conv_out_st and the final sigmoid
 * (model.py:92-96, 333-334, 372-373): a dot product per pixel, bandwidth-bound on the expanded tensor.  Exact fp32
 * whatever the GEMM precision of the plan; the summation order is a function of C only.
 * C % 256 == 0, C <= 2048; w9c tap-major as in uavsal_dw_desc; scale2 / bias2 point at ONE value each (device memory).
 * act: UAVSAL_ACT_NONE | UAVSAL_ACT_RELU6 | UAVSAL_ACT_SIGMOID.
 */
typedef struct uavsal_dw_dot_desc {
    const float* in;   int32_t ldi;                              /* [n_img, H, W, C] NHWC, row pitch ldi floats */
    const float* w9c;  const float* scale;  const float* bias;   /* depthwise taps [9][C], folded BN [C] */
    const float* w2;   const float* scale2; const float* bias2;  /* projection weights [C]; folded BN of its one output */
    float*       out;  int32_t ldo;                              /* [n_img * H * W] values, ldo floats apart */
    int32_t n_img, H, W, C, act;
} uavsal_dw_dot_desc;

int uavsal_dw3x3_dot(const uavsal_dw_dot_desc* d, uavsal_stream_t stream);

/*
 * Fused inverted-residual block: pw-expand + BN + ReLU6 -> depthwise 3x3 (stride 1 / 2, pad 1) + BN + ReLU6 ->
 * pw-linear + BN [+ x], ONE launch, the 6x-expanded tensors stay in LDS.  Replaces a whole torchvision
 * InvertedResidual / dwBlock (model.py:74-103, model_feature.py:62-66) where the block is bandwidth-bound:
 * instances exist for (Cin, hidden, Cout, stride) of MobileNetV2 features[1..7] -- (32,32,16,1) with w1 == NULL
 * (t = 1: no expand conv), (16,96,24,2), (24,144,24,1), (24,144,32,2), (32,192,32,1), (32,192,64,2);
 * uavsal_fused_ir_supported tells.  Exact fp32 FMA arithmetic whatever the GEMM precision of the plan.
 * Weights (host-packed, fp32): w1 [Cin][hidden] (= conv weight transposed), wd [9][hidden] tap-major as in
 * uavsal_dw_desc, w2 [hidden][Cout]; scale* / bias* are the folded BatchNorms.  res (or NULL) is the block
 * input for the residual connection (stride 1, Cin == Cout): out = bn3(...) + res.  in / out / res rows and every
 * per-channel vector are read and written 16 bytes at a time: 16-byte aligned bases, ld* multiples of 4.
 */
typedef struct uavsal_fused_ir_desc {
    const float* in;   int32_t ldi;
    const float* w1;   const float* scale1;  const float* bias1;      /* NULL for a block without expand conv */
    const float* wd;   const float* scale_d; const float* bias_d;
    const float* w2;   const float* scale2;  const float* bias2;
    const float* res;  int32_t ldr;
    float*       out;  int32_t ldo;
    int32_t n_img, H, W, Cin, hidden, Cout, stride;
    int32_t tile;                /* output patch per workgroup: 0 = by the launch size, 1 = 4x16, 2 = 16x16 (stride 1) / 8x16 (stride 2) */
} uavsal_fused_ir_desc;

int uavsal_fused_ir(const uavsal_fused_ir_desc* d, uavsal_stream_t stream);
/* 0: no instance.  1: an instance of the small-channel kernel (features[1..7]); w1 is [Cin][hidden], w2 [hidden][Cout]
 * (the 1x1 weights transposed).  2: an instance of the mid-channel kernel (csrc/fused_mid.hip: stride 1, (Cin, hidden, Cout)
 * in {(64,384,64), (64,384,96), (64,384,32), (96,576,96)}); w1 is [hidden][Cin], w2 [Cout][hidden] -- the conv weights' own
 * layout -- 16-byte aligned.  No launch. */
int uavsal_fused_ir_supported(const uavsal_fused_ir_desc* d);

/*
 * Stem: dense 3x3 stride-2 pad-1 convolution 3 -> 32 + BatchNorm + ReLU6 reading the
 * caller's NCHW fp32 frames and writing NHWC.  Replaces torchvision
 * mobilenet_v2().features[0] as run by ReMobileNetV2.forward (model_feature.py:63).
 * w is [27][32]: w[(ci*9 + ky*3 + kx)*32 + co] = weight[co, ci, ky, kx].
 * If `in_u8` is non-NULL it is used instead of `in`: uint8 RGB frames [n,3,H,W] that are
 * normalised on load, (v/255 - mean[c]) / stdv[c]  (utils_data.py:43-65 normalize_data).
 */
typedef struct uavsal_stem_desc {
    const float*   in;     /* [n_img, 3, H, W] fp32 NCHW, or NULL */
    const uint8_t* in_u8;  /* [n_img, 3, H, W] uint8, or NULL */
    const float* w;  const float* scale;  const float* bias;
    float* out;  int32_t ldo;            /* [n_img, Ho, Wo, 32] */
    int32_t n_img, H, W;
    float mean[3], stdv[3];
} uavsal_stem_desc;

int uavsal_stem_conv(const uavsal_stem_desc* d, uavsal_stream_t stream);

/*
 * Bilinear resize, align_corners=True, NHWC, writing into a channel slice.
 * Replaces F.interpolate(..., mode='bilinear', align_corners=True) at model.py:152-153
 * and :360, and -- through the source-image map -- `cb_cxt.repeat(time_dims,1,1,1)`
 * (model.py:361):  source image of output image n is (n % src_mod) / src_div.
 *   reference semantics (tiling quirk): src_mod = B, src_div = 1
 *   batched clips:                      src_mod = n_out, src_div = T
 *   plain resize:                       src_mod = n_out, src_div = 1
 */
typedef struct uavsal_bilinear_desc {
    const float* in;  int32_t ldi;  int32_t Hi, Wi;
    float* out;       int32_t ldo;  int32_t Ho, Wo;
    int32_t n_out, C, src_mod, src_div;
    void* out_split;  int32_t ldos;              /* optional split shadow, written IN ADDITION to `out` */
} uavsal_bilinear_desc;

int uavsal_bilinear_ac(const uavsal_bilinear_desc* d, uavsal_stream_t stream);

/*
 * Temporal neighbour differences of teConv_sub (model.py:194-200), per sequence of
 * `seq_len` consecutive images: out[:, 0:C] = x[t]-x[t-1] (t=0: x[1]-x[0]);
 * out[:, C:2C] = x[t]-x[t+1] (t=last: x[last-1]-x[last]).  seq_len >= 2.
 */
typedef struct uavsal_tdiff_desc {
    const float* in;  int32_t ldi;
    float* out;       int32_t ldo;
    int32_t n_img, HW, C, seq_len;
} uavsal_tdiff_desc;

int uavsal_tdiff(const uavsal_tdiff_desc* d, uavsal_stream_t stream);

/*
 * Winograd F(R x R, 3x3) transforms (R = 2 or 4, P = R + 2) of a dense 3x3 convolution with stride 1 and padding 1
 * (csrc/winograd.hip), fp32.
 *   uavsal_wino_input : `in` NHWC [n_img, H, W, C] (row stride ldi)  ->  `out` = V[P*P][Mp][ldo]: plane k = P i + j holds
 *                       (B^T d B)_{ij} of every R x R output tile (tile = (image, ty, tx), row-major), channels C.
 *   uavsal_wino_output: `in` = M[P*P][Mp][ldi] (the P*P GEMM results, channels C = Cout)  ->  `out` NHWC
 *                       [n_img, H, W, C]: y = A^T m A, then scale / bias (or NULL), ACT_NONE / ACT_RELU6, optional residual
 *                       (EPI_AFFINE), or the ConvTWA update h_t = g x_t + (1 - g) h_{t-1}, g = sigmoid(y + aux) with
 *                       res = x_t, hprev = h_{t-1} (EPI_TWA).
 * Mp (rows per plane) >= n_img * ceil(H/R) * ceil(W/R), a multiple of 128; rows past the tiles are not written.
 * The GEMM between them: uavsal_conv_gemm with a = V, n_img = P*P, H = Mp, W = 1, taps = 1, w = the P*P
 * transformed filter matrices (G g G^T)_k packed one after another, w_group_stride = their size in floats
 * (packing.pack_wino_weight).  Image strides are in pixels, 0 = H * W.
 */
typedef struct uavsal_wino_desc {
    const float* in;   int32_t ldi;  int64_t in_img_stride;
    float* out;        int32_t ldo;  int64_t out_img_stride;
    int32_t n_img, H, W, C;
    int64_t Mp;
    int32_t R;                   /* output tile: 2 = F(2x2, 3x3), 16 planes; 4 = F(4x4, 3x3), 36 planes */
    const float* scale; const float* bias;
    int32_t act, epi;
    const float* res;   int32_t ldr;  int64_t res_img_stride;
    const float* aux;   int32_t ldx;  int64_t aux_img_stride;
    const float* hprev; int32_t ldh;  int64_t h_img_stride;
    /* uavsal_wino_input only, n_seg = 1..3 (else 0 and `in` as above): the input is the channel concatenation of n_seg tensors,
     * seg_in[s] = [n_img][seg_H[s]][seg_W[s]][seg_ld[s]] NHWC with seg_c[s] channels (sum = C).  A segment on a map of another
     * size is resized to H x W on the fly, bilinear with align_corners=True, with uavsal_bilinear_ac's arithmetic -- the head of
     * the SRF-Net feeds conv_last with cat[interpolate(x5), interpolate(x4), lv3] (reference model.py:151-156): neither the
     * resized maps nor the concat buffer exist then.  `in`, `ldi`, `in_img_stride` are ignored. */
    int32_t n_seg;
    const float* seg_in[3];  int32_t seg_ld[3], seg_c[3], seg_H[3], seg_W[3];
} uavsal_wino_desc;

int uavsal_wino_input(const uavsal_wino_desc* d, uavsal_stream_t stream);
int uavsal_wino_output(const uavsal_wino_desc* d, uavsal_stream_t stream);

/* Sum over groups of T consecutive images (model.py:357-358): out[b] = sum_t in[b*T+t]. */
typedef struct uavsal_tsum_desc {
    const float* in;  int32_t ldi;
    float* out;       int32_t ldo;
    int32_t n_groups, T, HW, C;
} uavsal_tsum_desc;

int uavsal_tsum(const uavsal_tsum_desc* d, uavsal_stream_t stream);

/*
 * Layout change between the caller's NCHW tensors (priors cb[0], cb[1], recurrent
 * state; Demo_Test.py:14-27,85-86) and the NHWC working buffers.
 * to_nhwc != 0: nchw -> nhwc (channels C..Cpad-1 of the destination are zero-filled),
 * else nhwc -> nchw.  `ld` is the NHWC channel stride.
 */
typedef struct uavsal_layout_desc {
    const float* in;  float* out;
    int32_t n_img, C, HW, ld, to_nhwc, Cpad;
} uavsal_layout_desc;

int uavsal_layout(const uavsal_layout_desc* d, uavsal_stream_t stream);

/*
 * Output post-processing of the caller, replacing per frame
 * postprocess_predictions (utils_data.py:289-303: cv2.resize INTER_LINEAR to the source frame
 * size, centre crop, `img / max(img) * 255`) and np2mat/im2uint8 (utils_data.py:68-82: clip,
 * round half to even, uint8) as used at Demo_Test.py:89-91.
 * in: [n_img, h, w] fp32 maps; out: [n_img, H, W] uint8; scratch: >= 4*n_img bytes (zeroed here).
 */
typedef struct uavsal_post_desc {
    const float* in;  uint8_t* out;  void* scratch;
    int32_t n_img, h, w, H, W;
} uavsal_post_desc;

int uavsal_postprocess(const uavsal_post_desc* d, uavsal_stream_t stream);

/*
 * Strided row copy, device to device: `rows` rows of `row_floats` contiguous floats (a multiple of 4),
 * row r read at in + r*in_pitch and written at out + r*out_pitch (pitches in floats).  Used by the
 * persistent-state mode to carry h_last (NHWC, last frame of every clip of the ConvTWA history, which
 * IS the state: model_convlstm.py:377-381) into the resident state buffer that step 0 of the next call
 * reads -- the device-side form of `x_state = [out_state[0].detach()]` (Demo_Test.py:86).
 */
typedef struct uavsal_copy_desc {
    const float* in;  float* out;
    int64_t in_pitch, out_pitch, row_floats;  int32_t rows;
} uavsal_copy_desc;

int uavsal_copy_rows(const uavsal_copy_desc* d, uavsal_stream_t stream);

/*
 * Fill `n` 32-bit words at `out` with the bit pattern `bits`.  The activation arena's debug mode records one after
 * the last reader of every buffer (bits = a quiet NaN), so that a kernel that still reads a released range -- or a
 * later buffer placed over a live one -- shows up as NaN in the maps instead of as a plausible number
 * (the reference keeps O(group) memory per call, Demo_Test.py:75-86; here liveness does: engine.py).
 */
typedef struct uavsal_fill_desc {
    uint32_t* out;  int64_t n;  uint32_t bits;
} uavsal_fill_desc;

int uavsal_fill(const uavsal_fill_desc* d, uavsal_stream_t stream);

/*
 * Error guard: the last launch of a forward.  If `*err` is non-zero (a kernel of this run reported
 * UAVSAL_ERR_*), every listed buffer is overwritten with NaN, so that no caller can mistake the result
 * for a saliency map; `*err` is then copied to `*host_err` (a device-visible host word, may be NULL).
 * The caller-facing contract (SURVEY.md 8(b) "Errors: never silent") is completed on the host by
 * uavsal_plan_status.
 */
typedef struct uavsal_guard_desc {
    const int32_t* err;  int32_t* host_err;
    float* buf[3];  int64_t n[3];          /* fp32 buffers to poison; unused entries NULL / 0 */
} uavsal_guard_desc;

int uavsal_guard(const uavsal_guard_desc* d, uavsal_stream_t stream);

/* ---- launch plan: a recorded sequence of the calls above, run natively ------------ */
typedef struct uavsal_plan uavsal_plan;

uavsal_plan* uavsal_plan_create(void);
void uavsal_plan_destroy(uavsal_plan* p);
int uavsal_plan_add_conv(uavsal_plan* p, const uavsal_conv_desc* d);
int uavsal_plan_add_dw(uavsal_plan* p, const uavsal_dw_desc* d);
int uavsal_plan_add_dw_dot(uavsal_plan* p, const uavsal_dw_dot_desc* d);
int uavsal_plan_add_stem(uavsal_plan* p, const uavsal_stem_desc* d);
int uavsal_plan_add_bilinear(uavsal_plan* p, const uavsal_bilinear_desc* d);
int uavsal_plan_add_tdiff(uavsal_plan* p, const uavsal_tdiff_desc* d);
int uavsal_plan_add_wino_input(uavsal_plan* p, const uavsal_wino_desc* d);
int uavsal_plan_add_wino_output(uavsal_plan* p, const uavsal_wino_desc* d);
int uavsal_plan_add_tsum(uavsal_plan* p, const uavsal_tsum_desc* d);
int uavsal_plan_add_layout(uavsal_plan* p, const uavsal_layout_desc* d);
int uavsal_plan_add_copy(uavsal_plan* p, const uavsal_copy_desc* d);
int uavsal_plan_add_fused_ir(uavsal_plan* p, const uavsal_fused_ir_desc* d);
int uavsal_plan_add_fill(uavsal_plan* p, const uavsal_fill_desc* d);
/* The plan owns one device error word (for uavsal_conv_desc.err of its convs) and a host mirror of it.
 * add_guard records a uavsal_guard over up to three output buffers with those words filled in. */
int32_t* uavsal_plan_error_word(uavsal_plan* p);
int uavsal_plan_add_guard(uavsal_plan* p, float* b0, int64_t n0, float* b1, int64_t n1, float* b2, int64_t n2);
/* Outcome of the most recent uavsal_plan_run / uavsal_plan_graph_launch: 0 = completed without a device
 * error (or, with wait == 0, still running); UAVSAL_EDEVICE = a kernel set the error word (returned
 * once: the words are cleared; the caller re-zeroes its stream-K workspaces).  wait != 0 blocks until
 * the run has finished. */
int uavsal_plan_status(uavsal_plan* p, int wait);
/* Re-point one tensor of a recorded op at the caller's buffer, so that a forward reads the caller's frames /
 * priors / state in place and writes the map / state straight into freshly allocated outputs (no staging
 * copies, no clones; SURVEY.md 8(b): inputs are not modified, outputs are owned by the caller).
 * `slot`: conv 0 = a, 1 = out;  stem 0 = in (fp32), 1 = in_u8 (the other is cleared);  layout 0 = in, 1 = out;
 * guard 0..2 = buf[slot].  Shapes, strides and every other field stay as recorded.  Not available once the
 * plan has been captured into a graph (UAVSAL_ESTATE): a graph replays fixed addresses. */
int uavsal_plan_patch_ptr(uavsal_plan* p, int op, int slot, void* ptr);
/* Parallel branches: ops are recorded on the current lane (0 = the caller's stream, 1..7 = private
 * streams).  fork(l): lane l starts after everything recorded on lane 0 so far; join(l): lane 0 waits
 * for lane l.  Every forked lane must be joined before the plan ends.  A captured plan keeps the
 * branches as parallel graph nodes.  enable_lanes(0) runs everything on the caller's stream. */
int uavsal_plan_set_lane(uavsal_plan* p, int lane);
int uavsal_plan_add_fork(uavsal_plan* p, int lane);
int uavsal_plan_add_join(uavsal_plan* p, int lane);
int uavsal_plan_enable_lanes(uavsal_plan* p, int on);
int uavsal_plan_size(const uavsal_plan* p);
/* launch ops [first, last) in order on `stream` (last < 0: to the end).  Lanes are honoured when the whole plan runs; a sub-range is
 * launched flat on `stream`.  A run that reaches the end of the plan -- whole, or the second of two ranges (the streaming driver issues
 * everything in front of the recurrence, waits for the previous group's state, then the rest) -- is the one uavsal_plan_status reports on. */
int uavsal_plan_run(uavsal_plan* p, int first, int last, uavsal_stream_t stream);
/* capture the whole plan into a hipGraph once, then replay it */
int uavsal_plan_graph_build(uavsal_plan* p, uavsal_stream_t stream);
int uavsal_plan_graph_launch(uavsal_plan* p, uavsal_stream_t stream);

/* ---- device timing on the launch stream (hipEvent) and introspection --------------- */
/* run ops [first,last) `iters` times on `stream`, return average ms per iteration in *ms
 * (hipEventRecord on `stream` around the loop, hipEventSynchronize on the stop event). */
int uavsal_plan_time(uavsal_plan* p, int first, int last, int iters, uavsal_stream_t stream, float* ms);

int uavsal_abi_version(void);
int uavsal_sizeof_desc(int which); /* 0 conv,1 dw,2 stem,3 bilinear,4 tdiff,5 tsum,6 layout,7 post,8 guard,9 copy,10 fused_ir,11 wino,12 dw_dot,13 fill */
const char* uavsal_build_info(void);

#ifdef __cplusplus
}
#endif
#endif /* UAVSAL_HIP_H */

"""Single-operator wrappers over the C ABI (one launch each, outputs allocated through
torch, launched on torch's current stream).  Used by the per-kernel parity tests and
handy for experiments; the model path records plans instead (engine.py).
All activations are NHWC fp32 cuda tensors `[n, h, w, c]` (channel slices allowed:
pass a narrowed view `t[..., a:b]`)."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib as L
from . import packing as P


def _stream(t):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _nhwc_view(t: torch.Tensor):
    """(ptr, ld, n, h, w, c) of an NHWC tensor or of a channel slice of one."""
    if t.dim() != 4 or t.dtype != torch.float32 or not t.is_cuda:
        raise RuntimeError("expected a float32 cuda NHWC tensor [n,h,w,c]")
    n, h, w, c = t.shape
    # strides of size-1 dimensions are arbitrary in torch: take ld from the first real one
    ld = t.stride(2) if w > 1 else (t.stride(1) if h > 1 else (t.stride(0) if n > 1 else c))
    ok = (c == 1 or t.stride(3) == 1) and (w == 1 or t.stride(2) == ld) \
        and (h == 1 or t.stride(1) == w * ld) and (n == 1 or t.stride(0) == h * w * ld)
    if not ok:
        raise RuntimeError("tensor is not an NHWC buffer (or channel slice of one)")
    return t.data_ptr(), ld, n, h, w, c


def _streamk_outcome(ws, what):
    """After a synchronised stream-K launch: the error word (last int of the flag block when the descriptor
    carries no `err` pointer) must be clear and every flag reset."""
    flags = ws[:65536].view(torch.int32)
    if int(flags[-1].item()) != 0:
        ws[:65536].zero_()
        raise RuntimeError("%s: a stream-K hand-off timed out on the device (UAVSAL_ERR_STREAMK)" % what)
    if int(flags.abs().sum().item()) != 0:
        raise RuntimeError("stream-K workspace not clean after the launch")


def split_shadow(x: torch.Tensor) -> torch.Tensor:
    """Split shadow (include/uavsal_hip.h) of a dense fp32 NHWC tensor `[..., C]`, C % 32 == 0, computed with
    torch: fp16 `[..., C/32, 2, 32]` with [.., 0, :] = hi = fp16(16 x) and [.., 1, :] = lo = fp16(16 x - hi).
    (The kernels round hi toward zero; any hi works as long as lo is the residual.)"""
    c = x.shape[-1]
    if c % 32:
        raise RuntimeError("split shadows need C % 32 == 0")
    x16 = x.float().reshape(*x.shape[:-1], c // 32, 32) * 16.0
    hi = x16.clamp(-65504.0, 65504.0).to(torch.float16)
    lo = (x16 - hi.float()).clamp(-65504.0, 65504.0).to(torch.float16)
    return torch.stack([hi, lo], -2).contiguous()


def merge_shadow(sp: torch.Tensor) -> torch.Tensor:
    """fp32 `[..., C]` back from a split shadow `[..., C/32, 2, 32]`."""
    v = (sp[..., 0, :].float() + sp[..., 1, :].float()) / 16.0
    return v.reshape(*v.shape[:-2], v.shape[-2] * 32)


def _empty_shadow(n, h, w, c, device):
    if c % 32:
        raise RuntimeError("split shadows need C % 32 == 0")
    return torch.empty((n, h, w, c // 32, 2, 32), dtype=torch.float16, device=device)


def conv_gemm(x, weight, scale=None, bias=None, act=L.ACT_NONE, res=None, out=None, prec="f32", tile=0, dw=None,
              stream_k=False, sk_spin_limit=0, sk_debug_drop=0, split_in=False, split_out=False, n_group=0, _stale_streamk_flags=False):
    """Dense 1x1 / 3x3 conv (+ folded BN, activation, residual).  `weight` [Cout,Cin,k,k] (cpu or cuda).
    `n_group`: several 1x1 convs of different inputs as one launch -- `x` holds the inputs side by side ([.., groups * Cin]),
    `weight` [groups * n_group, Cin, 1, 1] the stacked weights (uavsal_conv_desc.n_group / a_group_off).
    `dw=(w[C,1,3,3], scale[C], bias[C], stride)`: x is the expanded tensor and the depthwise 3x3 + BN + ReLU6
    in front of this 1x1 conv is computed inside the GEMM's loader (fused inverted-residual tail)."""
    lib = L.load()
    ap, lda, n, hin, win, cin = _nhwc_view(x)
    stride = dw[3] if dw is not None else 1
    h, w = (hin - 1) // stride + 1, (win - 1) // stride + 1
    cout, taps = weight.shape[0], weight.shape[2] * weight.shape[3]
    if n_group:
        cin = weight.shape[1]
    if out is None:
        out = torch.empty((n, h, w, cout), dtype=torch.float32, device=x.device)
    op, ldc, *_ = _nhwc_view(out)
    keep = []
    d = L.ConvDesc()
    d.n_group, d.a_group_off = n_group, (cin if n_group else 0)
    d.a, d.lda, d.a_img_stride = ap, lda, hin * win
    if split_in:          # pre-split A operand: the GEMM stages it by LDS-DMA (f16x3, eligible shapes)
        if not x.is_contiguous():
            raise RuntimeError("split_in needs a dense NHWC tensor")
        xs = split_shadow(x)
        keep.append(xs)
        d.a_split, d.ldas = xs.data_ptr(), 2 * cin
    shadow = None
    if split_out:
        shadow = _empty_shadow(n, h, w, cout, x.device)
        d.out_split, d.ldos = shadow.data_ptr(), 2 * cout
    if dw is not None:
        w9 = P.pack_dw_weight(dw[0]).to(x.device)
        ds, db = dw[1].float().contiguous().to(x.device), dw[2].float().contiguous().to(x.device)
        keep += [w9, ds, db]
        d.dw_w9c, d.dw_scale, d.dw_bias = w9.data_ptr(), ds.data_ptr(), db.data_ptr()
        d.dw_stride, d.dw_Hin, d.dw_Win = stride, hin, win
    if scale is not None:
        npad = P.roundup(cout, 32)
        s = P.pad_vec(scale, npad, 1.0).to(x.device)
        b = P.pad_vec(bias, npad, 0.0).to(x.device)
        keep += [s, b]
        d.scale, d.bias = s.data_ptr(), b.data_ptr()
    d.out, d.ldc, d.o_img_stride = op, ldc, h * w
    if res is not None:
        rp, ldr, *_ = _nhwc_view(res)
        d.res, d.ldr, d.r_img_stride = rp, ldr, h * w
    d.n_img, d.H, d.W, d.Cin, d.Cout, d.taps = n, h, w, cin, cout, taps
    d.prec, d.act, d.epi, d.tile = L.PREC[prec], act, L.EPI_AFFINE, tile
    if stream_k or dw is not None:      # (the depthwise -> projection launch splits K of narrow outputs through it)
        ws = torch.zeros(int(lib.uavsal_streamk_workspace_bytes()), dtype=torch.uint8, device=x.device)
        keep.append(ws)
        d.sk_ws, d.sk_ws_bytes = ws.data_ptr(), ws.numel()
        d.sk_spin_limit, d.sk_debug_drop = sk_spin_limit, sk_debug_drop
        if _stale_streamk_flags:        # test hook: what a stream-K wait that gave up earlier on this lane leaves behind
            ws[:4 * L.SK_TICKET_BASE].view(torch.int32).fill_(1)
    d.w = 1 << 20
    uses_split = int(lib.uavsal_conv_uses_split(C.byref(d))) == 1
    if split_in and not uses_split:
        raise RuntimeError("this shape / tile does not take the pre-split path")
    dwproj = int(lib.uavsal_conv_dwproj(C.byref(d))) != 0
    k32 = prec == "f32" and int(lib.uavsal_conv_tile(C.byref(d))) in (8, 9, 10, 11)     # 32-float K stages: 3x3 K order differs
    wp = P.pack_conv_weight(weight, "f16x3i" if uses_split else ("f16x3j" if dwproj and prec == "f16x3" else
                                                                 ("f32k32" if k32 else prec))).to(x.device)
    keep.append(wp)
    d.w = wp.data_ptr()
    L.check(lib.uavsal_conv_gemm(C.byref(d), _stream(x)), "uavsal_conv_gemm")
    torch.cuda.current_stream(x.device).synchronize()   # `keep` must outlive the launch
    if stream_k and not _stale_streamk_flags:
        _streamk_outcome(ws, "uavsal_conv_gemm")
    return (out, shadow) if split_out else out


def conv3x3_winograd(x, weight, scale=None, bias=None, act=L.ACT_NONE, res=None, twa=None, r=2, size=None):
    """Dense 3x3 conv (stride 1, padding 1) through Winograd F(r x r, 3x3), r = 2 or 4, fp32: input transform, ONE GEMM
    launch over the (r + 2)^2 transform planes (per-plane weights), output transform with the epilogue.  `twa=(x_t, pre_t)`: `x` is
    h_{t-1} and the output transform applies the ConvTWA update (model_convlstm.py:276-292).
    `x` may be a list of up to three NHWC tensors: the conv's input is their channel concatenation, each first resized to
    `size = (h, w)` (bilinear, align_corners=True) when its map has another size -- inside the input transform
    (uavsal_wino_desc.n_seg)."""
    lib = L.load()
    segs = None
    if isinstance(x, (list, tuple)):
        segs = [_nhwc_view(t) for t in x]
        n, (h, w), cin = segs[0][2], size, sum(sg[5] for sg in segs)
        ip, ldi, x = None, 0, x[0]
    else:
        ip, ldi, n, h, w, cin = _nhwc_view(x)
    cout = weight.shape[0]
    tiles = n * ((h + r - 1) // r) * ((w + r - 1) // r)
    pp = (r + 2) * (r + 2)
    mp = P.roundup(tiles, 128)
    dev = x.device
    v = torch.zeros((pp, mp, cin), dtype=torch.float32, device=dev)
    m = torch.empty((pp, mp, cout), dtype=torch.float32, device=dev)
    out = torch.empty((n, h, w, cout), dtype=torch.float32, device=dev)
    wp = P.pack_wino_weight(weight, r).to(dev)
    st = _stream(x)
    wi = L.WinoDesc()
    wi.inp, wi.ldi, wi.out, wi.ldo = ip, ldi, v.data_ptr(), cin
    wi.n_img, wi.H, wi.W, wi.C, wi.Mp, wi.R = n, h, w, cin, mp, r
    if segs is not None:
        wi.n_seg = len(segs)
        for i, (sp_, sld_, sn_, sh_, sw_, sc_) in enumerate(segs):
            wi.seg_in[i], wi.seg_ld[i], wi.seg_c[i], wi.seg_H[i], wi.seg_W[i] = sp_, sld_, sc_, sh_, sw_
    L.check(lib.uavsal_wino_input(C.byref(wi), st), "uavsal_wino_input")
    d = L.ConvDesc()
    d.a, d.lda, d.a_img_stride = v.data_ptr(), cin, mp
    d.w = wp.data_ptr()
    d.w_group_stride = P.roundup(cout, 32) * P.roundup(cin, 32)
    d.out, d.ldc, d.o_img_stride = m.data_ptr(), cout, mp
    d.n_img, d.H, d.W, d.Cin, d.Cout, d.taps = pp, mp, 1, cin, cout, 1
    d.prec, d.act, d.epi, d.tile = L.PREC["f32"], L.ACT_NONE, L.EPI_AFFINE, 0
    L.check(lib.uavsal_conv_gemm(C.byref(d), st), "uavsal_conv_gemm(winograd planes)")
    wo = L.WinoDesc()
    wo.inp, wo.ldi, wo.out, wo.ldo = m.data_ptr(), cout, out.data_ptr(), cout
    wo.n_img, wo.H, wo.W, wo.C, wo.Mp, wo.R = n, h, w, cout, mp, r
    keep = [v, m, wp]
    if scale is not None:
        npad = P.roundup(cout, 32)
        s_ = P.pad_vec(scale, npad, 1.0).to(dev)
        b_ = P.pad_vec(bias, npad, 0.0).to(dev)
        keep += [s_, b_]
        wo.scale, wo.bias = s_.data_ptr(), b_.data_ptr()
    wo.act, wo.epi = act, L.EPI_AFFINE
    if res is not None:
        rp, ldr, *_ = _nhwc_view(res)
        wo.res, wo.ldr = rp, ldr
    if twa is not None:
        xp, ldr, *_ = _nhwc_view(twa[0])
        pp, ldx, *_ = _nhwc_view(twa[1])
        wo.epi, wo.res, wo.ldr, wo.aux, wo.ldx, wo.hprev, wo.ldh = L.EPI_TWA, xp, ldr, pp, ldx, ip, ldi
    L.check(lib.uavsal_wino_output(C.byref(wo), st), "uavsal_wino_output")
    torch.cuda.current_stream(dev).synchronize()
    return out


def twa_step(x_t, h_prev, pre_t, w_h, prec="f32", tile=0, stream_k=False):
    """One ConvTWA step given pre_t = conv3x3(W[:, :C], x_t): returns h_t (NHWC)."""
    lib = L.load()
    ap, lda, n, h, w, c = _nhwc_view(h_prev)
    xp, ldr, *_ = _nhwc_view(x_t)
    pp, ldx, *_ = _nhwc_view(pre_t)
    out = torch.empty((n, h, w, c), dtype=torch.float32, device=x_t.device)
    d = L.ConvDesc()
    d.a, d.lda, d.a_img_stride = ap, lda, h * w
    d.w = 1 << 20
    d.out, d.ldc, d.o_img_stride = out.data_ptr(), c, h * w
    d.res, d.ldr, d.r_img_stride = xp, ldr, h * w
    d.aux, d.ldx, d.x_img_stride = pp, ldx, h * w
    d.n_img, d.H, d.W, d.Cin, d.Cout, d.taps = n, h, w, c, c, 9
    d.prec, d.act, d.epi, d.tile = L.PREC[prec], L.ACT_NONE, L.EPI_TWA, tile
    if stream_k:
        ws = torch.zeros(int(lib.uavsal_streamk_workspace_bytes()), dtype=torch.uint8, device=x_t.device)
        d.sk_ws, d.sk_ws_bytes = ws.data_ptr(), ws.numel()
    # weights last: the fp32 kernels with 32-float K stages (tiles 8-10) take the 3x3 K order in 32-channel blocks
    k32 = prec == "f32" and int(lib.uavsal_conv_tile(C.byref(d))) in (8, 9, 10, 11)
    wp = P.pack_conv_weight(w_h, "f32k32" if k32 else prec).to(x_t.device)
    d.w = wp.data_ptr()
    L.check(lib.uavsal_conv_gemm(C.byref(d), _stream(x_t)), "uavsal_conv_gemm(TWA)")
    torch.cuda.current_stream(x_t.device).synchronize()
    if stream_k:
        _streamk_outcome(ws, "uavsal_conv_gemm(TWA)")
    return out


def dw3x3(x, weight, scale, bias, stride=1, dilation=1, act=L.ACT_RELU6, out=None, split_out=False, dil_groups=None):
    """`split_out`: the result is written as a split shadow (fp16 `[n,h,w,C/32,2,32]`) instead of fp32.
    `dil_groups`: a list of dilations, one per equal channel group (uavsal_dw_desc.dil_group_c)."""
    lib = L.load()
    ip, ldi, n, h, w, c = _nhwc_view(x)
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    if split_out:
        sp = _empty_shadow(n, ho, wo, c, x.device)
        op, ldo = None, c
    else:
        if out is None:
            out = torch.empty((n, ho, wo, c), dtype=torch.float32, device=x.device)
        op, ldo, *_ = _nhwc_view(out)
    w9 = P.pack_dw_weight(weight).to(x.device)
    s, b = scale.float().contiguous().to(x.device), bias.float().contiguous().to(x.device)
    d = L.DwDesc()
    d.inp, d.ldi, d.w9c, d.scale, d.bias = ip, ldi, w9.data_ptr(), s.data_ptr(), b.data_ptr()
    d.out, d.ldo = op, ldo
    d.n_img, d.H, d.W, d.C, d.stride, d.dilation, d.act = n, h, w, c, stride, dilation, act
    if split_out:
        d.out_split, d.ldos = sp.data_ptr(), 2 * c
    if dil_groups:
        d.dil_group_c = c // len(dil_groups)
        for i, dl in enumerate(dil_groups):
            d.dil_groups[i] = dl
    L.check(lib.uavsal_dw3x3(C.byref(d), _stream(x)), "uavsal_dw3x3")
    torch.cuda.current_stream(x.device).synchronize()
    return sp if split_out else out


def dw3x3_dot(x, weight, scale, bias, w2, scale2, bias2, act=L.ACT_NONE):
    """Depthwise 3x3 (stride 1) + BN + ReLU6 -> 1x1 projection to ONE channel + BN + `act` in one launch (uavsal_dw3x3_dot).
    `x` NHWC [n,h,w,C]; `weight` [C,1,3,3]; `w2` [1,C,1,1]; scale2 / bias2 one value each.  Returns [n,h,w,1]."""
    lib = L.load()
    ip, ldi, n, h, w, c = _nhwc_view(x)
    dev = x.device
    out = torch.empty((n, h, w, 1), dtype=torch.float32, device=dev)
    keep = [P.pack_dw_weight(weight).to(dev), scale.float().contiguous().to(dev), bias.float().contiguous().to(dev),
            w2.detach().float().reshape(-1).contiguous().to(dev), torch.as_tensor(scale2, dtype=torch.float32).reshape(1).to(dev),
            torch.as_tensor(bias2, dtype=torch.float32).reshape(1).to(dev)]
    d = L.DwDotDesc()
    d.inp, d.ldi = ip, ldi
    d.w9c, d.scale, d.bias, d.w2, d.scale2, d.bias2 = (t.data_ptr() for t in keep)
    d.out, d.ldo = out.data_ptr(), 1
    d.n_img, d.H, d.W, d.C, d.act = n, h, w, c, act
    L.check(lib.uavsal_dw3x3_dot(C.byref(d), _stream(x)), "uavsal_dw3x3_dot")
    torch.cuda.current_stream(dev).synchronize()
    return out


def fused_ir(x, w1, bn1, wd, bnd, w2, bn2, stride=1, residual=False, tile=0):
    """One inverted-residual block as a single launch (uavsal_fused_ir).  `x` NHWC; `w1` [hid,Cin,1,1] or None
    (no expand conv); `wd` [hid,1,3,3]; `w2` [Cout,hid,1,1]; bn* = (scale, bias) folded BatchNorms."""
    lib = L.load()
    ip, ldi, n, h, w, cin = _nhwc_view(x)
    hid, cout = wd.shape[0], w2.shape[0]
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    out = torch.empty((n, ho, wo, cout), dtype=torch.float32, device=x.device)
    dev = x.device
    keep = []

    def up(t):
        t = t.detach().float().contiguous().to(dev)
        keep.append(t)
        return t.data_ptr()
    d = L.FusedIrDesc()
    d.inp, d.ldi = ip, ldi
    d.n_img, d.H, d.W, d.Cin, d.hidden, d.Cout, d.stride, d.tile = n, h, w, cin, hid, cout, stride, tile
    d.w1 = (1 << 20) if w1 is not None else None
    kind = int(lib.uavsal_fused_ir_supported(C.byref(d)))
    if not kind:
        raise RuntimeError("no fused inverted-residual instance for (Cin, hidden, Cout, stride) = %s" % ((cin, hid, cout, stride),))
    natural = kind == 2          # the mid-channel kernel takes the 1x1 weights in their own layout, the small-channel one transposed
    if w1 is not None:
        d.w1 = up(w1.reshape(hid, cin) if natural else w1.reshape(hid, cin).t())
        d.scale1, d.bias1 = up(bn1[0]), up(bn1[1])
    d.wd, d.scale_d, d.bias_d = up(P.pack_dw_weight(wd)), up(bnd[0]), up(bnd[1])
    d.w2, d.scale2, d.bias2 = up(w2.reshape(cout, hid) if natural else w2.reshape(cout, hid).t()), up(bn2[0]), up(bn2[1])
    if residual:
        d.res, d.ldr = ip, ldi
    d.out, d.ldo = out.data_ptr(), cout
    L.check(lib.uavsal_fused_ir(C.byref(d), _stream(x)), "uavsal_fused_ir")
    torch.cuda.current_stream(dev).synchronize()
    return out


def stem_conv(x_nchw, weight, scale, bias):
    lib = L.load()
    n, _, H, W = x_nchw.shape
    ho, wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    out = torch.empty((n, ho, wo, 32), dtype=torch.float32, device=x_nchw.device)
    ws = P.pack_stem_weight(weight).to(x_nchw.device)
    s, b = scale.float().contiguous().to(x_nchw.device), bias.float().contiguous().to(x_nchw.device)
    x_nchw = x_nchw.contiguous()
    d = L.StemDesc()
    if x_nchw.dtype == torch.uint8:
        d.inp, d.in_u8 = None, x_nchw.data_ptr()
    else:
        d.inp, d.in_u8 = x_nchw.data_ptr(), None
    d.w, d.scale, d.bias, d.out, d.ldo = ws.data_ptr(), s.data_ptr(), b.data_ptr(), out.data_ptr(), 32
    d.n_img, d.H, d.W = n, H, W
    from .synth import IMAGENET_MEAN, IMAGENET_STD
    for i in range(3):
        d.mean[i], d.stdv[i] = IMAGENET_MEAN[i], IMAGENET_STD[i]
    L.check(lib.uavsal_stem_conv(C.byref(d), _stream(out)), "uavsal_stem_conv")
    torch.cuda.current_stream(out.device).synchronize()
    return out


def bilinear_ac(x, ho, wo, out=None, n_out=None, src_mod=None, src_div=1, split_out=False):
    lib = L.load()
    ip, ldi, n, h, w, c = _nhwc_view(x)
    n_out = n if n_out is None else n_out
    if out is None:
        out = torch.empty((n_out, ho, wo, c), dtype=torch.float32, device=x.device)
    op, ldo, *_ = _nhwc_view(out)
    d = L.BilinearDesc()
    d.inp, d.ldi, d.Hi, d.Wi, d.out, d.ldo, d.Ho, d.Wo = ip, ldi, h, w, op, ldo, ho, wo
    d.n_out, d.C, d.src_mod, d.src_div = n_out, c, (n_out if src_mod is None else src_mod), src_div
    if split_out:
        sp = _empty_shadow(n_out, ho, wo, c, x.device)
        d.out_split, d.ldos = sp.data_ptr(), 2 * c
    L.check(lib.uavsal_bilinear_ac(C.byref(d), _stream(x)), "uavsal_bilinear_ac")
    torch.cuda.current_stream(x.device).synchronize()
    return (out, sp) if split_out else out


def tdiff(x, seq_len):
    lib = L.load()
    ip, ldi, n, h, w, c = _nhwc_view(x)
    out = torch.empty((n, h, w, 2 * c), dtype=torch.float32, device=x.device)
    d = L.TdiffDesc()
    d.inp, d.ldi, d.out, d.ldo, d.n_img, d.HW, d.C, d.seq_len = ip, ldi, out.data_ptr(), 2 * c, n, h * w, c, seq_len
    L.check(lib.uavsal_tdiff(C.byref(d), _stream(x)), "uavsal_tdiff")
    return out


def tsum(x, T):
    lib = L.load()
    ip, ldi, n, h, w, c = _nhwc_view(x)
    out = torch.empty((n // T, h, w, c), dtype=torch.float32, device=x.device)
    d = L.TsumDesc()
    d.inp, d.ldi, d.out, d.ldo, d.n_groups, d.T, d.HW, d.C = ip, ldi, out.data_ptr(), c, n // T, T, h * w, c
    L.check(lib.uavsal_tsum(C.byref(d), _stream(x)), "uavsal_tsum")
    return out


def to_nhwc(x_nchw, cpad=0):
    lib = L.load()
    x_nchw = x_nchw.contiguous()
    n, c, h, w = x_nchw.shape
    ld = max(c, cpad)
    out = torch.empty((n, h, w, ld), dtype=torch.float32, device=x_nchw.device)
    d = L.LayoutDesc()
    d.inp, d.out, d.n_img, d.C, d.HW, d.ld, d.to_nhwc, d.Cpad = x_nchw.data_ptr(), out.data_ptr(), n, c, h * w, ld, 1, cpad
    L.check(lib.uavsal_layout(C.byref(d), _stream(out)), "uavsal_layout")
    return out


def to_nchw(x_nhwc):
    lib = L.load()
    ip, ld, n, h, w, c = _nhwc_view(x_nhwc)
    out = torch.empty((n, c, h, w), dtype=torch.float32, device=x_nhwc.device)
    d = L.LayoutDesc()
    d.inp, d.out, d.n_img, d.C, d.HW, d.ld, d.to_nhwc, d.Cpad = ip, out.data_ptr(), n, c, h * w, ld, 0, 0
    L.check(lib.uavsal_layout(C.byref(d), _stream(out)), "uavsal_layout")
    return out


def postprocess_predictions(maps, H, W):
    """Device version of the caller's post-processing (Demo_Test.py:89-91): `maps` [n,1,h,w] or
    [n,h,w] fp32 cuda -> uint8 [n,H,W] cuda (resize to the frame size, crop, /max*255, rint)."""
    lib = L.load()
    if maps.dim() == 4:
        maps = maps[:, 0]
    maps = maps.contiguous()
    if maps.dtype != torch.float32 or not maps.is_cuda:
        raise RuntimeError("expected float32 cuda maps")
    n, h, w = maps.shape
    out = torch.empty((n, H, W), dtype=torch.uint8, device=maps.device)
    scratch = torch.empty((n,), dtype=torch.int32, device=maps.device)
    d = L.PostDesc()
    d.inp, d.out, d.scratch = maps.data_ptr(), out.data_ptr(), scratch.data_ptr()
    d.n_img, d.h, d.w, d.H, d.W = n, h, w, H, W
    L.check(lib.uavsal_postprocess(C.byref(d), _stream(maps)), "uavsal_postprocess")
    torch.cuda.current_stream(maps.device).synchronize()
    return out


def lstm_step(x_t, h_prev, c_prev, weight, prec="f32"):
    """One ConvLSTM step (reference model_convlstm.py:111-126, bias=False) from NHWC tensors and the
    reference-layout weight [4*hid, in+hid, 3, 3]: returns (h_t, c_t).  The x half of the conv is
    hoisted exactly as the engine does (`aux`), gates interleaved as n = 4*c + gate."""
    lib = L.load()
    ap, lda, n, h, w, hid = _nhwc_view(h_prev)
    cin = x_t.shape[3]
    wi = weight.reshape(4, hid, cin + hid, 3, 3).permute(1, 0, 2, 3, 4).reshape(4 * hid, cin + hid, 3, 3)
    pre = conv_gemm(x_t, wi[:, :cin].contiguous(), None, None, prec=prec)       # [n,h,w,4*hid]
    cp, ldr, *_ = _nhwc_view(c_prev)
    h_out = torch.empty((n, h, w, hid), dtype=torch.float32, device=x_t.device)
    c_out = torch.empty_like(h_out)
    wp = P.pack_conv_weight(wi[:, cin:].contiguous(), prec).to(x_t.device)
    d = L.ConvDesc()
    d.a, d.lda, d.a_img_stride = ap, lda, h * w
    d.w = wp.data_ptr()
    d.out, d.ldc, d.o_img_stride = h_out.data_ptr(), hid, h * w
    d.out2, d.ld2 = c_out.data_ptr(), hid
    d.res, d.ldr, d.r_img_stride = cp, ldr, h * w
    d.aux, d.ldx, d.x_img_stride = pre.data_ptr(), 4 * hid, h * w
    d.n_img, d.H, d.W, d.Cin, d.Cout, d.taps = n, h, w, hid, 4 * hid, 9
    d.prec, d.act, d.epi, d.tile = L.PREC[prec], L.ACT_NONE, L.EPI_LSTM, 0
    L.check(lib.uavsal_conv_gemm(C.byref(d), _stream(x_t)), "uavsal_conv_gemm(LSTM)")
    torch.cuda.current_stream(x_t.device).synchronize()
    return h_out, c_out

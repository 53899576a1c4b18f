"""ORACLE (test infrastructure, not product code): CPU fp32 restatement of the
reference UAVSal per-frame inference forward.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this file.  The product path (`iip_uavsal_saliency_amd`) never does; it
fails loudly when the HIP library is missing.

Parity pin: `oracle/make_goldens.py` runs the reference's own `model.py` /
`model_convlstm.py` (unmodified, from /root/reference, with a torchvision-free
stand-in for `model_feature`) on the synthetic weights/inputs of
`iip_uavsal_saliency_amd.synth` and commits the outputs under `tests/golden/`.
`tests/test_oracle_golden.py` checks this restatement against those vectors.
The MobileNetV2 backbone arithmetic lives in torchvision (0.5.0 / 0.8.2 per the
reference README.md:27,34), which is absent here: it is restated from its
published definition and pinned structurally by the reference's known answer of
51.59 MB parameters+buffers (Tools/Getmodelsize_demo.py:93).

Every function cites the reference lines it follows (paths relative to
/root/reference).  All arithmetic is torch CPU fp32 (`F.conv2d`,
`F.batch_norm`, `F.interpolate`), i.e. the same ATen CPU operators the reference
executes.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

# (expand t, out channels c, repeats n, first stride s) -- torchvision MobileNetV2
# inverted_residual_setting; reference taps it at features[0:2],[2:4],[4:7],[7:14],[14:18]
# (model_feature.py:62-69).
MBV2_SETTING = ((1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2),
                (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1))
MBV2_TAPS = (2, 4, 7, 14, 18)


def _cbr(cin, cout, k=1, stride=1, dilation=1, groups=1):
    """conv(no bias)+BN+ReLU6 triple, keys `.0/.1` (model.py:65-72 BasicConv2d;
    torchvision ConvBNReLU has the same layout)."""
    pad = dilation * (k - 1) // 2
    return nn.Sequential(
        nn.Conv2d(cin, cout, k, stride, pad, dilation=dilation, groups=groups, bias=False),
        nn.BatchNorm2d(cout),
        nn.ReLU6(inplace=False))


class _IRBlock(nn.Module):
    """Inverted residual: `.conv` = [pw-expand cbr]? + dw cbr + pw-linear conv + BN.
    model.py:74-103 (dwBlock) and torchvision InvertedResidual share this layout."""

    def __init__(self, cin, cout, stride=1, expand=6, dilation=1, res_connect=None):
        super().__init__()
        hid = int(round(cin * expand))
        self.residual = (stride == 1 and cin == cout)
        if res_connect is not None:                      # model.py:82-84
            self.residual = bool(res_connect) and self.residual
        seq = []
        if expand != 1:
            seq.append(_cbr(cin, hid, 1))
        seq += [_cbr(hid, hid, 3, stride, dilation, groups=hid),
                nn.Conv2d(hid, cout, 1, 1, 0, bias=False), nn.BatchNorm2d(cout)]
        self.conv = nn.Sequential(*seq)

    def forward(self, x):
        y = self.conv(x)
        return x + y if self.residual else y


class _Backbone(nn.Module):
    """`.features` = torchvision mobilenet_v2().features (19 entries; [18] is held
    but never run, model_feature.py:68)."""

    def __init__(self):
        super().__init__()
        layers: List[nn.Module] = [_cbr(3, 32, 3, stride=2)]
        cin = 32
        for t, c, n, s in MBV2_SETTING:
            for i in range(n):
                layers.append(_IRBlock(cin, c, s if i == 0 else 1, expand=t))
                cin = c
        layers.append(_cbr(cin, 1280, 1))
        self.features = nn.Sequential(*layers)

    def forward(self, x):
        outs = []
        lo = 0
        for hi in MBV2_TAPS:                              # model_feature.py:63-67
            x = self.features[lo:hi](x)
            outs.append(x)
            lo = hi
        return outs


class _SRFNet(nn.Module):
    """model.py:110-158 (uavsal_srfnet_aspp)."""

    def __init__(self, planes=256):
        super().__init__()
        self.conv_lv3 = _cbr(32, 64, 1)
        self.conv_lv4 = _cbr(96, 128, 1)
        self.lv5_aspp1 = _cbr(320, 256, 1)
        self.lv5_aspp2 = _IRBlock(320, 256, dilation=6)
        self.lv5_aspp3 = _IRBlock(320, 256, dilation=12)
        self.lv5_aspp4 = _IRBlock(320, 256, dilation=18)
        self.conv_lv5 = _cbr(1024, 256, 1)
        self.conv_last = _cbr(448, planes, 3)
        self.features = _Backbone()

    def forward(self, x, taps=None):
        _, _, c3, c4, c5 = self.features(x)
        a = torch.cat([self.lv5_aspp1(c5), self.lv5_aspp2(c5),
                       self.lv5_aspp3(c5), self.lv5_aspp4(c5)], 1)      # model.py:142-146
        x5 = self.conv_lv5(a)
        x4 = self.conv_lv4(c4)
        x3 = self.conv_lv3(c3)
        size = c3.shape[2:]
        x5 = F.interpolate(x5, size=size, mode="bilinear", align_corners=True)  # :152
        x4 = F.interpolate(x4, size=size, mode="bilinear", align_corners=True)  # :153
        out = self.conv_last(torch.cat([x5, x4, x3], 1))                        # :155-156
        if taps is not None:
            taps.update(c3=c3, c4=c4, c5=c5, aspp=a, sfnet=out)
        return out


class _SpConv(nn.Module):                                   # model.py:163-171
    def __init__(self, planes):
        super().__init__()
        self.spconv = _IRBlock(planes, planes, res_connect=False)

    def forward(self, x):
        return self.spconv(x)


def temporal_differences(x1: torch.Tensor) -> torch.Tensor:
    """model.py:194-200.  Frame i gets cat[x1[i]-x1[i-1], x1[i]-x1[i+1]]; frame 0 gets
    cat[x1[1]-x1[0], x1[0]-x1[1]]; the last frame cat[x1[-1]-x1[-2], x1[-2]-x1[-1]].
    Needs >= 2 frames (the reference raises on 1)."""
    n = x1.shape[0]
    if n < 2:
        raise RuntimeError("teConv_sub needs at least 2 frames per call (reference model.py:194)")
    prev = torch.empty_like(x1)
    nxt = torch.empty_like(x1)
    prev[1:] = x1[1:] - x1[:-1]
    prev[0] = x1[1] - x1[0]
    nxt[:-1] = x1[:-1] - x1[1:]
    nxt[-1] = x1[-2] - x1[-1]
    return torch.cat([prev, nxt], 1)


class _TeConv(nn.Module):                                   # model.py:173-208
    def __init__(self, planes, reduction):
        super().__init__()
        width = planes // reduction
        self.reduce_conv = _cbr(planes, width, 1)
        self.sub_conv = _IRBlock(2 * width, width, res_connect=False)
        self.last_conv = _cbr(width, planes, 1)

    def forward(self, x):
        return self.last_conv(self.sub_conv(temporal_differences(self.reduce_conv(x))))


class _STBlock(nn.Module):                                  # model.py:210-249, fu_type='sum'
    def __init__(self, planes, reduction):
        super().__init__()
        self.stconv_sp = _SpConv(planes)
        self.stconv_te = _TeConv(planes, reduction)
        self.stconv_last = _cbr(planes, planes, 1)

    def forward(self, x):
        return x + self.stconv_last(self.stconv_sp(x) + self.stconv_te(x))


class _TWACell(nn.Module):                                  # model_convlstm.py:238-295
    def __init__(self, cin, hid):
        super().__init__()
        self.rnn_conv = nn.Conv2d(cin + hid, hid, 3, padding=1, bias=False)

    def forward(self, x_t, h):
        i = torch.sigmoid(self.rnn_conv(torch.cat([x_t, h], 1)))   # :279-283
        return i * x_t + (1 - i) * h                                # :290


class _TWA(nn.Module):                                      # model_convlstm.py:297-401
    def __init__(self, cin, hid):
        super().__init__()
        self.cell_list = nn.ModuleList([_TWACell(cin, hid)])

    def forward(self, seq, h):
        """seq `[b, t, c, h, w]`, h `[b, c, h, w]` -> (all h_t stacked on dim 1, h_last)."""
        outs = []
        for t in range(seq.shape[1]):                               # :368-371
            h = self.cell_list[0](seq[:, t], h)
            outs.append(h)
        return torch.stack(outs, 1), h


class _LSTMCell(nn.Module):                                 # model_convlstm.py:73-130
    def __init__(self, cin, hid):
        super().__init__()
        self.rnn_conv = nn.Conv2d(cin + hid, 4 * hid, 3, padding=1, bias=False)


class _LSTM(nn.Module):                                     # model_convlstm.py:132-236
    def __init__(self, cin, hid):
        super().__init__()
        self.cell_list = nn.ModuleList([_LSTMCell(cin, hid)])

    def forward(self, seq, hc):
        h, c = hc
        outs = []
        for t in range(seq.shape[1]):                               # :209-213
            h, c = convlstm_cell_step(self.cell_list[0].rnn_conv.weight, seq[:, t], h, c)
            outs.append(h)
        return torch.stack(outs, 1), (h, c)


class RefUAVSal(nn.Module):
    """model.py:254-375 (UAVSal), `cnn_type='mobilenet_v2'`, any 0/1 `bias_type` (default `[1,1,1]`: 685 entries).
    Same attribute names / state_dict keys as the reference; a disabled prior has no module (model.py:288-324)."""

    def __init__(self, time_dims=5, num_stblock=2, planes=256, rnn="twa", bias_type=(1, 1, 1)):
        super().__init__()
        self.time_dims = time_dims
        self.use_gauss_prior, self.use_ob_prior, self.use_context_prior = (int(b) for b in bias_type)   # model.py:281-283
        self.num_cb = sum(1 for b in bias_type if b > 0)                                                 # :284
        self.rnn_type = rnn      # "lstm" = the reference's UAVSAL_LSTM (model.py:960-1076)
        self.sfnet = _SRFNet(planes)
        self.st_layer = nn.Sequential(*[_STBlock(planes, planes // 32) for _ in range(num_stblock)])
        self.fust_layer = nn.Sequential(_IRBlock(planes, planes))
        if self.use_gauss_prior:                                                                         # :289-296
            self.gauss_cb_layer = nn.Sequential(_IRBlock(8, 64), _IRBlock(64, 64))
        if self.use_ob_prior:                                                                            # :298-305
            self.ob_cb_layer = nn.Sequential(_IRBlock(20, 64), _IRBlock(64, 64))
        if self.use_context_prior:                                                                       # :307-314
            self.cxt_cb_prior = nn.Sequential(_IRBlock(planes, 64, stride=2), _IRBlock(64, 64, stride=2))
        if self.num_cb:                                                                                  # :316-324
            self.fucb_layer = nn.Sequential(_IRBlock(64 * self.num_cb, planes // 4))
            self.fucbst_layer = nn.Sequential(_IRBlock(planes + planes // 4, planes))
        self.rnn = _LSTM(planes, planes) if rnn == "lstm" else _TWA(planes, planes)
        self.conv_out_st = _IRBlock(planes, 1)

    @torch.no_grad()
    def forward(self, x, cb, in_state=None, taps: Optional[Dict[str, torch.Tensor]] = None):
        """x `[B*T,3,H,W]`, cb=[gauss `[B*T,8,h,w]`, ob `[B*T,20,h,w]`], in_state=None or
        [`[1,256,h,w]`] -> (out `[B*T,1,h,w]`, [h_last]).  `None` state = zeros on x's
        device (the reference calls `.cuda()`, model_convlstm.py:294-295)."""
        T = self.time_dims
        x = self.sfnet(x, taps)
        for i, blk in enumerate(self.st_layer):
            x = blk(x)
            if taps is not None:
                taps[f"st{i}"] = x
        x = self.fust_layer(x)
        n, c, h, w = x.shape
        if self.num_cb:                                                  # model.py:346
            cb_fu = []
            if self.use_gauss_prior:
                cb_fu.append(self.gauss_cb_layer(cb[0]))                 # :349
            if self.use_ob_prior:
                cb_fu.append(self.ob_cb_layer(cb[1]))                    # :352
            if self.use_context_prior:
                B = n // T
                ctx = x.contiguous().view(B, T, c, h, w).sum(1)         # :357-358
                ctx = self.cxt_cb_prior(ctx)
                ctx = F.interpolate(ctx, size=(h, w), mode="bilinear", align_corners=True)
                cb_fu.append(ctx.repeat(T, 1, 1, 1))                     # :361 (tiles, not interleaves)
            x_cb = self.fucb_layer(torch.cat(cb_fu, 1))                  # :363-364
            x = self.fucbst_layer(torch.cat([x, x_cb], 1))               # :365
            if taps is not None:
                taps.update(fust_in_cb=x_cb)
        if taps is not None:
            taps.update(prefuse=x)
        if self.rnn_type == "lstm":      # in_state = None or [(h, c)]; returns [h, c]
            hc = in_state[0] if in_state is not None else (x.new_zeros(1, c, h, w), x.new_zeros(1, c, h, w))
            seq, (h_last, c_last) = self.rnn(x.view(1, n, c, h, w), hc)
            x = seq.reshape(n, c, h, w)
            logits = self.conv_out_st(x)
            if taps is not None:
                taps.update(rnn=x, logits=logits)
            return torch.sigmoid(logits), [h_last, c_last]
        h0 = in_state[0] if in_state is not None else x.new_zeros(1, c, h, w)
        seq, h_last = self.rnn(x.view(1, n, c, h, w), h0)            # :367-369
        x = seq.reshape(n, c, h, w)
        logits = self.conv_out_st(x)                                 # :372
        if taps is not None:
            taps.update(rnn=x, logits=logits)
        return torch.sigmoid(logits), [h_last]                       # :373-375

    @torch.no_grad()
    def forward_clips(self, x, cb, states=None, taps=None):
        """Batched-clips semantics of SURVEY.md 8(a): `x [C,T,3,H,W]`, cb=[`[C,T,8,h,w]`,
        `[C,T,20,h,w]`], states `[C,256,h,w]` or None -> (out `[C,T,1,h,w]`, states
        `[C,256,h,w]`) == C independent reference calls with time_dims=T."""
        C, T = x.shape[:2]
        saved = self.time_dims
        self.time_dims = T
        outs, sts = [], []
        try:
            for c in range(C):
                st = None if states is None else [states[c:c + 1]]
                o, s = self.forward(x[c], [None if t is None else t[c] for t in cb], st)
                outs.append(o)
                sts.append(s[0])
        finally:
            self.time_dims = saved
        return torch.stack(outs, 0), torch.cat(sts, 0)


def convlstm_cell_step(weight: torch.Tensor, x_t: torch.Tensor, h: torch.Tensor, c: torch.Tensor):
    """model_convlstm.py:111-126 (ConvLSTMCell.forward), bias=False: gates i,f,o,g from one
    3x3 conv of cat[x,h]; c' = f*c + i*g; h' = o*tanh(c')."""
    hid = h.shape[1]
    cc = F.conv2d(torch.cat([x_t, h], 1), weight, padding=1)
    ci, cf, co, cg = torch.split(cc, hid, dim=1)
    i, f, o, g = torch.sigmoid(ci), torch.sigmoid(cf), torch.sigmoid(co), torch.tanh(cg)
    c_next = f * c + i * g
    return o * torch.tanh(c_next), c_next


def build_oracle(time_dims=5, seed=0, rnn="twa", bias_type=(1, 1, 1)) -> RefUAVSal:
    """Oracle model in eval mode with the deterministic synthetic weights."""
    from iip_uavsal_saliency_amd import synth
    m = RefUAVSal(time_dims=time_dims, rnn=rnn, bias_type=bias_type)
    synth.load_synth_weights(m, seed)
    return m.eval()

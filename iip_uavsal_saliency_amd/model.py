"""Drop-in `UAVSal(nn.Module)` for the reference's per-frame saliency inference path.

Same constructor, attribute names, `state_dict` keys (685 entries, 13 407 338
parameters) and `forward(x, cb, in_state) -> (out, x_state)` contract as the
reference's `model.UAVSal` (reference model.py:254-375), so `Demo_Test.py:33-35,85`
works unchanged -- but the module tree only *holds parameters*: every
convolution, BatchNorm, ReLU6, concat, resize, temporal difference, the ConvTWA
recurrence and the final sigmoid execute as hand-written HIP kernels for gfx950
through `engine.Engine` (C ABI: include/uavsal_hip.h).  There is no eager /
CPU fallback: calling `forward` with CPU tensors or without the built library
raises.

Extension beyond the reference surface: `forward_clips` (batched independent
clips) and the `precision` attribute selecting how fp32 data is
fed to the matrix cores ('f32' exact fp32 MFMA -- the default and the reference's own
precision; 'f16x3' split-fp16 with fp32 accumulation, ~21 mantissa bits, <= 5e-4 on every
golden; 'bf16x3' / 'bf16' exist as diagnostics only and do not meet 1e-3).
"""
from __future__ import annotations

import os
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn

from .model_convlstm import ConvLSTM, ConvTWA
from .model_feature import ReMobileNetV2, _no_eager

feature_inplanes = {"mobilenet_v2": [24, 32, 96, 320]}       # reference model.py:32


def init_weights(model, funcname="kaiming_normal", **kwargs):
    """Random init with the reference's rules (model.py:49-60): convs by `funcname`,
    BatchNorm weight 1 / bias 0."""
    fn = {"kaiming_normal": nn.init.kaiming_normal_, "kaiming_uniform": nn.init.kaiming_uniform_,
          "xavier_uniform": nn.init.xavier_uniform_, "xavier_normal": nn.init.xavier_normal_,
          "normal": nn.init.normal_, "uniform": nn.init.uniform_}[funcname]
    for m in model.modules():
        if isinstance(m, nn.Conv2d):
            fn(m.weight, **kwargs)
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.ones_(m.weight)
            nn.init.zeros_(m.bias)


class BasicConv2d(nn.Sequential):
    """Conv2d(bias=False) + BatchNorm2d + ReLU6 (reference model.py:65-72)."""

    def __init__(self, in_planes, out_planes, kernel_size=3, stride=1, dilation=1, groups=1):
        padding = dilation * (kernel_size - 1) // 2
        super().__init__(
            nn.Conv2d(in_planes, out_planes, kernel_size, stride, padding, dilation=dilation, groups=groups, bias=False),
            nn.BatchNorm2d(out_planes), nn.ReLU6(inplace=True))

    def forward(self, x):
        _no_eager("BasicConv2d")


class dwBlock(nn.Module):
    """Inverted residual block (reference model.py:74-103)."""

    def __init__(self, inp, oup, kernel_size=3, stride=1, expand_ratio=6, dilation=1, res_connect=None):
        super().__init__()
        assert stride in (1, 2) and kernel_size == 3
        hidden = int(round(inp * expand_ratio))
        self.stride, self.dilation, self.expand_ratio = stride, dilation, expand_ratio
        self.cin, self.cout, self.hidden = inp, oup, hidden
        self.use_res_connect = stride == 1 and inp == oup
        if res_connect is not None:
            self.use_res_connect = bool(res_connect) and self.use_res_connect
        layers = []
        if expand_ratio != 1:
            layers.append(BasicConv2d(inp, hidden, kernel_size=1))
        layers += [BasicConv2d(hidden, hidden, kernel_size, stride=stride, dilation=dilation, groups=hidden),
                   nn.Conv2d(hidden, oup, 1, 1, 0, bias=False), nn.BatchNorm2d(oup)]
        self.conv = nn.Sequential(*layers)

    def forward(self, x):
        _no_eager("dwBlock")


class uavsal_srfnet_aspp(nn.Module):
    """SRF-Net (reference model.py:110-158)."""

    def __init__(self, cnn_type="mobilenet_v2", planes=(64, 64, 128, 256), last_channel=256):
        super().__init__()
        if cnn_type.lower() != "mobilenet_v2":
            raise NotImplementedError("the HIP path covers the mobilenet_v2 backbone (Demo_Test.py:33)")
        planes = list(planes)
        if last_channel == 128:
            planes = [32, 32, 64, 128]
        inpl = feature_inplanes["mobilenet_v2"]
        self.conv_lv3 = BasicConv2d(inpl[1], planes[1], 1)
        self.conv_lv4 = BasicConv2d(inpl[2], planes[2], 1)
        self.lv5_aspp1 = BasicConv2d(inpl[3], planes[3], 1)
        self.lv5_aspp2 = dwBlock(inpl[3], planes[3], 3, dilation=6)
        self.lv5_aspp3 = dwBlock(inpl[3], planes[3], 3, dilation=12)
        self.lv5_aspp4 = dwBlock(inpl[3], planes[3], 3, dilation=18)
        self.conv_lv5 = BasicConv2d(planes[3] * 4, planes[3], 1)
        self.conv_last = BasicConv2d(planes[1] + planes[2] + planes[3], last_channel, 3)
        init_weights(self, "kaiming_normal", mode="fan_out")
        self.features = ReMobileNetV2(name="mobilenet_v2")

    def forward(self, x):
        _no_eager("uavsal_srfnet_aspp")


class spConv(nn.Module):
    def __init__(self, inplanes, planes=256, kernel_size=3, stride=1, expand_ratio=6, dilation=1, res_connect=False):
        super().__init__()
        self.spconv = dwBlock(inplanes, planes, kernel_size, stride, expand_ratio, dilation, res_connect)
        init_weights(self, "kaiming_normal", mode="fan_out")

    def forward(self, x):
        _no_eager("spConv")


class teConv_sub(nn.Module):
    def __init__(self, inplanes, planes=256, time_dims=8, reduction=8, res_connect=False):
        super().__init__()
        self.time_dims = time_dims
        self.res_connect = res_connect and inplanes == planes
        width = planes // reduction
        self.reduce_conv = BasicConv2d(inplanes, width, 1)
        self.sub_conv = dwBlock(2 * width, width, 3, res_connect=False)
        self.last_conv = BasicConv2d(width, planes, 1)
        init_weights(self, "kaiming_normal", mode="fan_out")

    def forward(self, x):
        _no_eager("teConv_sub")


class STBlock(nn.Module):
    def __init__(self, inplanes, planes=256, time_dims=8, fu_type="sum", res_connect=True, **kwargs):
        super().__init__()
        if fu_type.lower() != "sum":
            raise NotImplementedError("UAVSal builds STBlock with fu_type='sum' (model.py:274)")
        self.res_connect = res_connect and inplanes == planes
        self.time_dims, self.fu_type, self.inplanes, self.planes = time_dims, "sum", inplanes, planes
        self.stconv_sp = spConv(inplanes, planes, res_connect=False)
        self.stconv_te = teConv_sub(inplanes, planes, time_dims, res_connect=False, **kwargs)
        self.stconv_last = BasicConv2d(planes, planes, 1)
        init_weights(self, "kaiming_normal", mode="fan_out")

    def forward(self, x):
        _no_eager("STBlock")


class UAVSal(nn.Module):
    """See module docstring.  Constructor arguments as reference model.py:255-261."""
    rnn_type = "twa"

    def __init__(self, cnn_type="mobilenet_v2", time_dims=5, num_stblock=2, bias_type=[1, 1, 1],
                 iosize=[360, 640, 45, 80], planes=256, pre_model_path="", precision="f32"):
        super().__init__()
        bias_type = [int(b) for b in bias_type]
        if len(bias_type) != 3 or any(b not in (0, 1) for b in bias_type):
            # (the reference sizes fucb_layer as 64 * sum(bias_type) but concatenates 64 channels per enabled prior,
            # model.py:316-318, 346-363: any other value fails there in its first forward)
            raise ValueError("bias_type must be three 0/1 flags [gauss, observed, context] (reference model.py:281-284)")
        if planes != 256:
            raise NotImplementedError("planes=256 is the only configuration on the Demo_Test path")
        self.time_dims = time_dims
        self.precision = precision
        self.use_graph = False          # replay the launch plan as one hipGraph
        self.fuse_dw = None             # None: engine default (fp32: LDS-halo fused dw->projection on the big blocks; engine.py)
        self.use_lanes = True           # independent branches on parallel streams / graph branches
        self.stream_k = True            # fp32 GEMMs: split K across workgroups when whole tiles leave CUs idle
        self.presplit = None            # f16x3: producers also write hi/lo fp16 shadows, GEMMs stage them by LDS-DMA;
                                        # None = from four clips up (measured: 17.71 vs 18.00 ms at eight clips, 3.240 vs 3.217 at one)
        self.fuse_blocks = True         # features[1..7]: whole inverted-residual block in one launch (uavsal_fused_ir)
        self.winograd = True            # exact-fp32 mode: dense 3x3 convs as Winograd F(4x4 / 2x2, 3x3) (csrc/winograd.hip)
        self.prec_overrides = None      # diagnostics: {op-name prefix: precision} for single layers (engine.Engine._prec_for)
        self.winograd_r = None          # None: F(4x4) for the all-frames convs, steps F(2x2) below four clips / F(4x4) from four up
                                        # (so the fp32 result depends on how clips are batched, ~1e-4); 2: F(2x2) everywhere (strict)
        # A device-side error (a stream-K hand-off that timed out) always NaN-fills the returned map and state.
        # True: the call also waits for its own launches (one event wait) and raises before returning;
        # False: fully asynchronous, the RuntimeError comes from the next call / `check_errors()`.
        # None (default): `forward` -- the reference surface, whose caller reads the map right away
        # (Demo_Test.py:87) -- waits; `forward_clips` -- the throughput surface -- does not: the wait costs the
        # host/GPU overlap between calls, ~3 % of a 5.8 ms step (profiles/r2_step_timeline.md).
        self.sync_errors = None
        # Opt-in persistent recurrent state (SURVEY.md 8(b) "Ownership", BASELINE configs[4]): the state stays
        # in the engine's NHWC buffer between calls; the returned state is a channels-last VIEW of that buffer
        # (valid until the next call overwrites it) and passing it back costs nothing.  Default False keeps the
        # reference's ownership rule: fresh tensors, never aliased (Demo_Test.py:86).
        self.persistent_state = False
        # Priors handed over as ONE map set broadcast over the frames (zero frame stride: what `priors.get_bias` returns -- the
        # reference's caller repeats one prior file over all frames, utils_data.py:466-467, 601-602): the two prior nets run on
        # one frame instead of on every frame.  Decided from the strides, never from the values; False: always per frame.
        self.dedupe_priors = True
        # Activations of a plan live in ONE arena per plan, placed by liveness (first / last use over the recorded launches; a
        # use on a side lane counts from the lane's fork to its join): memory per call follows the largest set of tensors that
        # is live at once, not the number of layers (720x1280, 4 x 16 frames: 14.7 GB instead of 42.6).  False: one allocation
        # per activation for the life of the plan (rounds 1-4).  `arena_debug`: every range is NaN-filled right after its last
        # declared use, so that a use after release shows up as NaN maps (tests).
        self.arena = True
        self.arena_debug = False
        self.max_engines = 4            # launch plans kept per model (LRU); packed weights are shared by all
        self.check_weight_versions = True   # rebuild plans when a parameter/buffer was modified in place
        self.sfnet = uavsal_srfnet_aspp(cnn_type, last_channel=planes)
        self.num_stblock = num_stblock
        self.st_layer = nn.Sequential(*[
            STBlock(planes, planes, time_dims=time_dims, reduction=planes // 32, res_connect=True)
            for _ in range(num_stblock)])
        self.fust_layer = nn.Sequential(dwBlock(planes, planes, kernel_size=3))
        self.use_gauss_prior, self.use_ob_prior, self.use_context_prior = bias_type
        self.num_cb = int(np.sum(np.array(bias_type) > 0))
        # every prior net exists only when its flag is set, and the two fusion blocks only when any is (reference
        # model.py:288-324): a disabled prior has no parameters and no state_dict keys
        if self.use_gauss_prior:
            self.gauss_cb_layer = nn.Sequential(dwBlock(8, 64, kernel_size=3), dwBlock(64, 64, kernel_size=3))
            init_weights(self.gauss_cb_layer)
        if self.use_ob_prior:
            self.ob_cb_layer = nn.Sequential(dwBlock(20, 64, kernel_size=3), dwBlock(64, 64, kernel_size=3))
            init_weights(self.ob_cb_layer)
        if self.use_context_prior:
            self.cxt_cb_prior = nn.Sequential(dwBlock(planes, 64, kernel_size=3, stride=2),
                                              dwBlock(64, 64, kernel_size=3, stride=2))
            init_weights(self.cxt_cb_prior)
        if self.num_cb:
            self.fucb_layer = nn.Sequential(dwBlock(64 * self.num_cb, planes // 4, kernel_size=3))
            self.fucbst_layer = nn.Sequential(dwBlock(planes + planes // 4, planes, kernel_size=3))
        _, _, shape_r_out, shape_c_out = iosize
        rnn_cls = ConvLSTM if self.rnn_type == "lstm" else ConvTWA
        self.rnn = rnn_cls((shape_r_out, shape_c_out), planes, planes, kernel_size=(3, 3), num_layers=1,
                           batch_first=True, bias=False, return_all_layers=False)
        self.conv_out_st = dwBlock(planes, 1, kernel_size=3)
        init_weights(self.st_layer, "kaiming_normal", mode="fan_out")
        init_weights(self.fust_layer, "kaiming_normal", mode="fan_out")
        init_weights(self.conv_out_st, "kaiming_normal", mode="fan_out")
        self._engines: "OrderedDict[tuple, object]" = OrderedDict()
        self._wshared: Dict[str, dict] = {}          # packed device weights per device, shared by its engines
        self._wversion = None
        if pre_model_path and os.path.exists(pre_model_path):
            # the reference's checkpoints are whole pickled models (model.py:339): resolved through the shim
            from .checkpoint import load_reference_state_dict
            self.load_state_dict(load_reference_state_dict(pre_model_path), strict=False)

    # -- engines are built from the current parameter values; drop them when those change
    def replica(self):
        """A second handle on the SAME parameters (and packed device weights) with its own launch plans and
        buffers.  One model's calls are ordered on its lanes; requests of independent videos issued through
        different replicas on different host streams overlap on the GPU and fill the phases a single forward leaves
        part of the chip idle in (the ConvTWA steps, the small backbone maps): `stream.RequestPipeline`,
        tools/pipeline_probe.py.  Settings are copied as they are now."""
        import copy
        r = copy.copy(self)                      # parameters, modules and settings by reference; no runtime state (__getstate__)
        r._wshared, r._wversion = self._wshared, self._wversion
        if "_wtensors" in self.__dict__:
            r._wtensors = self._wtensors
        return r

    _RUNTIME_STATE = ("_engines", "_wshared", "_wversion", "_wtensors", "_stream_replicas", "_stream_streams")

    def __getstate__(self):
        """`torch.save(model)` (the form of the reference's checkpoints, model.py:339), `pickle` and `copy.deepcopy` carry the
        parameters, buffers and settings -- never launch plans, packed device weights or streams, which are handles of this
        process; the copy packs and records its own at its first call."""
        return {k: v for k, v in self.__dict__.items() if k not in self._RUNTIME_STATE}

    def __setstate__(self, state):
        super().__setstate__(state)
        self._engines = OrderedDict()
        self._wshared = {}
        self._wversion = None

    def _drop_engines(self):
        self._engines = OrderedDict()
        self._wshared = {}
        self._wversion = None

    invalidate_engines = _drop_engines

    def _weights_version(self):
        return sum(t._version for t in self._wtensors)

    def check_errors(self):
        """Wait for every launched forward and raise if one reported a device-side error."""
        for eng in list(self._engines.values()):
            eng.check(wait=True)

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self._drop_engines()
        return out

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._drop_engines()
        return out

    def train(self, mode: bool = True):
        if mode:
            self._drop_engines()
        return super().train(mode)

    @staticmethod
    def frame_invariant(cb, frame_dims=1):
        """Are the caller's priors ONE map set for every frame -- structurally: every frame dimension of both tensors is a
        broadcast (stride 0, e.g. `maps[None].expand(n, -1, -1, -1)`, which is what `priors.get_bias` returns) or has one entry?
        Never decided from the values: materialised copies (`np.repeat`, `.repeat`, `.contiguous()`) take the general plan."""
        return all(all(t.shape[d] == 1 or t.stride(d) == 0 for d in range(frame_dims)) for t in cb)

    def _check_weights(self):
        """Drops the plans and the packed device weights when a parameter or buffer was modified since they were packed."""
        if not self.check_weight_versions:
            return
        if self._wversion is not None and self._weights_version() != self._wversion:
            # in-place ops on the parameters (no_grad), optimizer steps
            # (edits through `param.data` do NOT bump the version counter: call invalidate_engines() after those)
            self._drop_engines()
        if self._wversion is None:
            self._wtensors = [t for k, t in self.state_dict(keep_vars=True).items()
                              if not k.endswith("num_batches_tracked")]
            self._wversion = self._weights_version()

    def _engine(self, device, n_seq, seq_len, H, W, ctx_mode, taps=False, in_dtype=torch.float32, sync_default=True, static_priors=False):
        from .engine import Engine
        self._check_weights()
        key = (str(device), n_seq, seq_len, H, W, self.time_dims if ctx_mode == "tile" else seq_len,
               ctx_mode, self.precision, bool(taps), in_dtype, bool(self.use_graph), self.rnn_type, self.fuse_dw, bool(self.use_lanes),
               bool(self.stream_k), bool(self.persistent_state), tuple(getattr(self, "_sk_debug", (0, 0))),
               self.presplit, bool(self.fuse_blocks), bool(getattr(self, "winograd", True)), getattr(self, "winograd_r", None),
               tuple(sorted((getattr(self, "prec_overrides", None) or {}).items())), bool(static_priors),
               bool(getattr(self, "arena", True)), bool(getattr(self, "arena_debug", False)))
        eng = self._engines.get(key)
        if eng is None:
            while len(self._engines) >= max(1, int(self.max_engines)):
                _, old = self._engines.popitem(last=False)       # least recently used
                old.check(wait=True)
            eng = Engine(self, device, n_seq=n_seq, seq_len=seq_len, H=H, W=W,
                         ctx_T=key[5], ctx_mode=ctx_mode, precision=self.precision, taps=taps,
                         in_dtype=in_dtype, use_graph=self.use_graph, fuse_dw=self.fuse_dw,
                         use_lanes=self.use_lanes, stream_k=self.stream_k, persistent=self.persistent_state,
                         wcache=self._wshared.setdefault(str(torch.device(device)), {}), static_priors=static_priors)
            self._engines[key] = eng
        else:
            self._engines.move_to_end(key)
        eng.sync_errors = sync_default if self.sync_errors is None else bool(self.sync_errors)
        return eng

    def _check_common(self, x):
        if self.training:
            raise RuntimeError("UAVSal HIP path is inference-only: call model.eval() (Demo_Test.py:55)")
        if not x.is_cuda:
            raise RuntimeError("UAVSal.forward needs tensors on the MI355X (cuda) device; "
                               "there is no CPU fallback in this package")
        if x.dtype != torch.float32 and x.dtype != torch.uint8:
            raise RuntimeError("frames must be float32 (ImageNet-normalised) or uint8 RGB")

    @torch.no_grad()
    def forward(self, x, cb, in_state=None, taps: Optional[dict] = None):
        """x `[B*T,3,H,W]` float32 NCHW (ImageNet-normalised RGB; uint8 RGB is also accepted and
        normalised on load), cb = [gauss `[B*T,8,h,w]`, ob `[B*T,20,h,w]`], in_state = None or
        [`[1,256,h,w]`]  ->  (out `[B*T,1,h,w]`, [h_last `[1,256,h,w]`]); reference model.py:341-375.
        The frames of one call form ONE sequence for the temporal differences and the recurrence;
        the context prior is summed per chunk of `time_dims` frames and tiled (model.py:357-361)."""
        self._check_common(x)
        if x.dim() != 4 or x.shape[1] != 3:
            raise RuntimeError("x must be [B*T, 3, H, W]")
        n, _, H, W = x.shape
        if n < 2:
            raise RuntimeError("UAVSal.forward needs at least 2 frames per call (reference teConv_sub, model.py:194)")
        if n % self.time_dims:
            raise RuntimeError("shape '[%d, %d, ...]' is invalid for input of %d frames (model.py:357)" % (
                n // self.time_dims, self.time_dims, n))
        cb0, cb1 = self._used_cb(cb)
        static = self.dedupe_priors and n > 1 and self._static(cb0, cb1, 1)
        eng = self._engine(x.device, 1, n, H, W, "tile", taps is not None, x.dtype, static_priors=static)
        h, w = eng.h, eng.w
        self._check_cb(cb0, cb1, n, h, w)
        st, cst = None, None
        if in_state is not None:
            st = in_state[0]
            if self.rnn_type == "lstm":           # reference ConvLSTM: hidden_state[0] = (h, c)
                st, cst = st
            if tuple(st.shape) != (1, 256, h, w) or (cst is not None and tuple(cst.shape) != (1, 256, h, w)):
                raise RuntimeError("in_state tensors must be [1, 256, %d, %d]" % (h, w))
            if st.dtype != torch.float32 or st.device != x.device:
                raise RuntimeError("in_state must be float32 on the frames' device")
        out, state = eng.run(x, cb0, cb1, st, taps, cstate=cst)
        if self.rnn_type == "lstm":               # reference returns last_state_list[-1] = [h, c]
            return out.view(n, 1, h, w), [state[0].view(1, 256, h, w), state[1].view(1, 256, h, w)]
        return out.view(n, 1, h, w), [state.view(1, 256, h, w)]

    @torch.no_grad()
    def forward_clips(self, x, cb, states=None, taps: Optional[dict] = None):
        """Batched independent clips: x `[C,T,3,H,W]`, cb = [`[C,T,8,h,w]`,
        `[C,T,20,h,w]`], states `[C,256,h,w]` or None -> (out `[C,T,1,h,w]`, states `[C,256,h,w]`),
        equal to C reference calls with time_dims=T and a zero (or the given) state each."""
        self._check_common(x)
        if x.dim() != 5 or x.shape[2] != 3:
            raise RuntimeError("x must be [C, T, 3, H, W]")
        C, T, _, H, W = x.shape
        if T < 2:
            raise RuntimeError("each clip needs at least 2 frames (reference teConv_sub, model.py:194)")
        cb0, cb1 = self._used_cb(cb)
        for t in (cb0, cb1):
            if t is not None and (t.dim() != 5 or tuple(t.shape[:2]) != (C, T)):
                raise RuntimeError("cb tensors must be [C, T, channels, h, w]")
        static = self.dedupe_priors and self._static(cb0, cb1, 2)
        eng = self._engine(x.device, C, T, H, W, "clip", taps is not None, x.dtype, sync_default=False, static_priors=static)
        h, w = eng.h, eng.w
        cb0 = None if cb0 is None else cb0.reshape(C * T, *cb0.shape[2:])
        cb1 = None if cb1 is None else cb1.reshape(C * T, *cb1.shape[2:])
        self._check_cb(cb0, cb1, C * T, h, w)
        cst = None
        if self.rnn_type == "lstm" and states is not None:
            states, cst = states                 # (h [C,256,h,w], c [C,256,h,w])
        if states is not None and tuple(states.shape) != (C, 256, h, w):
            raise RuntimeError("states must be [C, 256, h, w]")
        out, state = eng.run(x.reshape(C * T, 3, H, W), cb0, cb1, states, taps, cstate=cst)
        if self.rnn_type == "lstm":
            return out.view(C, T, 1, h, w), (state[0].view(C, 256, h, w), state[1].view(C, 256, h, w))
        return out.view(C, T, 1, h, w), state.view(C, 256, h, w)

    def _used_cb(self, cb):
        """The caller's prior tensors this model reads: `cb[0]` iff the gaussian prior net exists, `cb[1]` iff the observed one
        does (reference model.py:348-353 indexes `cb` exactly so; entries of disabled priors are never touched)."""
        need = 2 if self.use_ob_prior else (1 if self.use_gauss_prior else 0)
        if need and (cb is None or len(cb) < need):
            raise RuntimeError("cb must be [gauss priors, observed priors] (Demo_Test.py:14-27)")
        return (cb[0] if self.use_gauss_prior else None), (cb[1] if self.use_ob_prior else None)

    def _static(self, cb0, cb1, frame_dims):
        used = [t for t in (cb0, cb1) if t is not None]
        return bool(used) and all(t.dim() > frame_dims for t in used) and self.frame_invariant(used, frame_dims)

    @staticmethod
    def _check_cb(cb0, cb1, n, h, w):
        for t, c, what in ((cb0, 8, "gauss"), (cb1, 20, "observed")):
            if t is None:
                continue
            if tuple(t.shape) != (n, c, h, w):
                raise RuntimeError("%s priors must be [%d,%d,%d,%d], got %s" % (what, n, c, h, w, tuple(t.shape)))
            if t.dtype != torch.float32:
                raise RuntimeError("cb tensors must be float32")


class UAVSAL_LSTM(UAVSal):
    """The reference's ConvLSTM ablation model (reference model.py:960-1076): identical to UAVSal
    except `self.rnn = ConvLSTM(...)` (model.py:1029); `in_state` is None or `[(h, c)]`, the returned
    state is `[h, c]` (model_convlstm.py:206-222)."""
    rnn_type = "lstm"

"""Video-level driver: the loop of the reference's `test()` (Demo_Test.py:65-95) with frames,
recurrent state and outputs kept on the device.  The reference decodes a video with cv2,
normalises each group of `batch_size * time_dims` frames on the host, copies it to the GPU, runs
`model(x, cb, state)`, copies the maps back and resizes them one by one with cv2.  Here the
caller hands over the uint8 RGB frames as a tensor (decoding is out of scope: no video codec in
this image); normalisation happens inside the stem kernel, the state never leaves HBM, and the
maps are resized / normalised / quantised by `uavsal_postprocess`."""
from __future__ import annotations

import math
import os
from typing import Optional

import torch

from . import matio, ops


def save_salmap(path: str, sal_u8: torch.Tensor, save_frames: Optional[int] = None) -> None:
    """The result file of the reference's loop (Demo_Test.py:92-95): `salmap` uint8 `[H, W, 1, F]` as MATLAB v7.3, the
    first `save_frames` frames (`saveFrames`, Demo_Test.py:92).  `sal_u8`: uint8 `[F, H, W]` (what `predict_video` returns)."""
    a = sal_u8.detach().cpu().numpy()
    if a.dtype.name != "uint8" or a.ndim != 3:
        raise ValueError("save_salmap takes the uint8 [F, H, W] maps of predict_video")
    if save_frames is not None:
        a = a[:max(0, int(save_frames))]
    matio.savemat(path, {"salmap": a[:, :, :, None].transpose(1, 2, 3, 0)})


def _host_streams(dev, n):
    """`n` host streams that really run side by side.  The HIP runtime multiplexes the streams of one priority onto
    GPU_MAX_HW_QUEUES (default 4) hardware queues, a stream taking the least used one when it is created; two streams on one
    queue execute strictly in submission order.  With the plans' lane streams and torch's pool around, two ordinary streams can
    end up on one queue, and the overlap is gone without a word (measured: 2077 frames/s -> 1851, below the 1888 of the plain
    loop; profiles/r5_experiments.md).  High-priority streams draw from a queue pool of their own, which nothing else in this
    package uses; consecutive ones get different queues."""
    prio = int(os.environ.get("UAVSAL_HOST_STREAM_PRIORITY", "-1"))
    return [torch.cuda.Stream(dev, priority=prio) for _ in range(n)]


_CACHE_KEYS = ("_stream_replicas", "_stream_streams", "_stream_copy")


class _Groups:
    """The groups of a video as device tensors.  Frames already on the device are sliced; frames in host memory (what a decoder
    hands over, Demo_Test.py:78-85) are uploaded group by group on a copy stream of their own, `ahead` groups in front of the
    one being launched, so that the copy of group k + 1 runs under the launches of group k instead of in front of the whole
    video (pinned memory: asynchronous; pageable memory: the host thread stages it, the GPU keeps computing)."""

    def __init__(self, model, frames_u8, group, steps, dev, ahead=2):
        self.frames, self.group, self.steps, self.dev, self.ahead = frames_u8, group, steps, dev, ahead
        self.host = not frames_u8.is_cuda
        self.pending = {}
        if self.host:
            cs = model.__dict__.get("_stream_copy")
            if cs is None or cs.device != torch.device(dev):
                cs = model.__dict__["_stream_copy"] = _host_streams(dev, 1)[0]
            self.copy_stream = cs

    def _fetch(self, i):
        if i < self.steps and i not in self.pending:
            with torch.cuda.stream(self.copy_stream):
                t = self.frames[i * self.group:(i + 1) * self.group].to(self.dev, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self.copy_stream)
            self.pending[i] = (t, ev)

    def get(self, i, stream=None):
        """Group `i`, ready on `stream` (default: the current one)."""
        if not self.host:
            return self.frames[i * self.group:(i + 1) * self.group]
        for k in range(i, i + 1 + self.ahead):
            self._fetch(k)
        t, ev = self.pending.pop(i)
        stream = stream or torch.cuda.current_stream(self.dev)
        stream.wait_event(ev)
        t.record_stream(stream)
        return t


def _inflight_replicas(model, n):
    """`n` handles on `model`'s weights for forwards that overlap on the GPU, cached on the model.  They run WITHOUT lanes: a
    plan's lanes are worth +0.8 % by themselves (fp32, one clip), but with two plans in flight their five streams each compete
    with the host streams for four hardware queues, and the overlap varies between 1884 and 2059 frames/s with the order things
    were created in; lane-less plans give 2075-2087 in every order (profiles/r5_experiments.md).  Same kernels, same order per
    buffer: the maps do not change."""
    model._check_weights()                    # (the model itself may never run: its handles must not inherit stale weights)
    reps = model.__dict__.get("_stream_replicas") or []
    while len(reps) < n:
        reps.append(model.replica())
    for rep in reps:
        engines = rep._engines if rep.__dict__.get("_wshared") is model.__dict__.get("_wshared") else None
        rep.__dict__.update({k: v for k, v in model.__dict__.items() if k not in _CACHE_KEYS})
        rep._engines = engines if engines is not None else type(model._engines)()
        rep.use_lanes = False
        for k in _CACHE_KEYS:
            rep.__dict__.pop(k, None)
    model.__dict__["_stream_replicas"] = reps
    return reps[:n]


@torch.no_grad()
def _predict_overlapped(model, groups, gauss_prior, ob_prior, steps, dev):
    """The groups of ONE video two deep in flight: group k runs on replica k % 2 and host stream k % 2; everything in front of
    the recurrence -- backbone, SRF-Net, ST blocks, prior fusion, the hoisted half of the gate convolution: 3.5 of a group's 4.2 ms
    at 8 frames -- does not depend on the previous group and is launched at once; the recurrence waits for the previous group's
    last launch and takes over its state (`Engine.run_streamed`).  The maps are those of the sequential loop, bit for bit."""
    # the two handles are kept on the model between videos (their launch plans cost ~50 ms to build); they follow the model's
    # current settings, and lose their plans when the model's packed weights were dropped (load_state_dict, in-place edits)
    models = _inflight_replicas(model, 2)
    streams = model.__dict__.get("_stream_streams")
    if streams is None or streams[0].device != torch.device(dev):
        streams = model.__dict__["_stream_streams"] = _host_streams(dev, 2)
    caller = torch.cuda.current_stream(dev)
    maps, prev_eng, prev_done = [], None, None
    for s_ in streams:
        s_.wait_stream(caller)                    # the frames / priors were produced on the caller's stream
    for i in range(steps):
        m_, s_ = models[i % 2], streams[i % 2]
        x = groups.get(i, s_)
        n = x.shape[0]
        with torch.cuda.stream(s_):
            cb = [gauss_prior.unsqueeze(0).expand(n, -1, -1, -1), ob_prior.unsqueeze(0).expand(n, -1, -1, -1)]
            cb0, cb1 = m_._used_cb(cb)
            static = m_.dedupe_priors and m_._static(cb0, cb1, 1)
            eng = m_._engine(dev, 1, n, x.shape[2], x.shape[3], "tile", False, x.dtype, static_priors=static)
            m_._check_cb(cb0, cb1, n, eng.h, eng.w)
            out = eng.run_streamed(x, cb0, cb1, prev=prev_eng, prev_done=prev_done, reset=(i == 0))
            prev_done = torch.cuda.Event()
            prev_done.record(s_)
            prev_eng = eng
            out.record_stream(caller)             # read on the caller's stream below
            maps.append(out.view(n, 1, eng.h, eng.w))
    for s_ in streams:
        caller.wait_stream(s_)
    for m_ in models:
        m_.check_errors()
    lstm = getattr(model, "rnn_type", "twa") == "lstm"
    return maps, ([(prev_eng.h_view, prev_eng.c_view)] if lstm else [prev_eng.h_view])


@torch.no_grad()
def predict_video(model, frames_u8: torch.Tensor, gauss_prior: torch.Tensor, ob_prior: torch.Tensor,
                  batch_size: int = 4, out_size: Optional[tuple] = None, return_maps: bool = False,
                  persistent_state: bool = True, out_path: Optional[str] = None, save_frames: Optional[int] = None,
                  overlap: Optional[bool] = None):
    """`frames_u8` uint8 `[F,3,H,W]` RGB (already letterboxed to the model size, as
    preprocess_videos does, utils_data.py:255-287) on the device, or in host memory (pinned for asynchronous copies): host
    frames are uploaded group by group on a copy stream, two groups ahead of the launches (`_Groups`), `gauss_prior` `[8,h,w]`, `ob_prior` `[20,h,w]`
    float32 (one map set, repeated per frame like get_bias, Demo_Test.py:14-27).
    Frames beyond the last full `time_dims` chunk are dropped (Demo_Test.py:68-70); groups of
    `batch_size * time_dims` frames are pushed through `model.forward` with the state carried
    (Demo_Test.py:75-86).  Returns uint8 `[F', H_out, W_out]` on the device (the reference's
    `pred_mat[..., 0]`), and the raw maps if asked; with `out_path` the maps are also written as the reference's
    `salmap` `[H,W,1,F]` v7.3 .mat file (Demo_Test.py:93-95).
    `overlap`: consecutive groups two deep in flight on two replicas -- only the recurrence of a group waits for the previous
    group (`_predict_overlapped`); same maps, bit for bit, 1833 -> 2006 frames/s on a 192-frame video at 360x640 in groups of 8.
    None (default): whenever it applies -- resident state, launch-loop plans, at least two whole groups (a shorter last group
    follows them on its own plan, taking over the state); True: insist (raises where it does not apply); False: the reference's
    one-after-the-other loop."""
    dev = next(model.parameters()).device
    T = model.time_dims
    F = frames_u8.shape[0]
    count_bs = F // T
    keep = count_bs * T
    if keep < 2:
        raise RuntimeError("need at least one full chunk of time_dims >= 2 frames")
    frames_u8 = frames_u8[:keep]
    if frames_u8.is_cuda and frames_u8.device != torch.device(dev):
        frames_u8 = frames_u8.to(dev)
    H, W = frames_u8.shape[2:]
    out_size = out_size or (H, W)
    group = batch_size * T
    steps = math.ceil(count_bs / batch_size)
    state = None
    maps = []
    was = model.persistent_state
    model.persistent_state = bool(persistent_state)
    try:
        whole = frames_u8.shape[0] // group              # a shorter last group runs on a plan of its own, after the others
        applies = bool(persistent_state) and not model.use_graph and whole >= 2
        if overlap is None:
            overlap = applies
        if overlap and not applies:
            raise RuntimeError("overlap=True needs persistent_state=True, the launch-loop plan and at least two whole groups of "
                               "batch_size * time_dims frames")
        groups = _Groups(model, frames_u8, group, steps, dev)
        first = 0
        if overlap:
            maps, state = _predict_overlapped(model, groups, gauss_prior.to(dev), ob_prior.to(dev), whole, dev)
            first = whole
        for i in range(first, steps):
            x = groups.get(i)
            n = x.shape[0]
            # one map set for every frame, handed over as a zero-stride view: the model runs its prior nets once per call
            # (model.dedupe_priors) instead of once per frame
            cb = [gauss_prior.to(dev).unsqueeze(0).expand(n, -1, -1, -1), ob_prior.to(dev).unsqueeze(0).expand(n, -1, -1, -1)]
            out, st = model(x, cb, state)
            # persistent mode: st[0] is a view of the engine's state buffer (valid until the next call, which
            # recognises it by address); a shorter last group runs on another plan, which loads it as a tensor
            # (the ConvLSTM variant carries (h, c): reference model_convlstm.py:204)
            state = [(st[0].detach(), st[1].detach())] if getattr(model, "rnn_type", "twa") == "lstm" else [st[0].detach()]
            maps.append(out)
    finally:
        model.persistent_state = was
    maps = torch.cat(maps, 0)
    sal = ops.postprocess_predictions(maps, out_size[0], out_size[1])
    if out_path is not None:
        save_salmap(out_path, sal, save_frames)
    return (sal, maps) if return_maps else sal


class RequestPipeline:
    """Independent requests (clips of DIFFERENT videos: no carried state between them) kept `streams` deep in
    flight: request k runs on host stream k % streams through its own handle on the model (`_inflight_replicas`: same weights,
    own lane-less launch plans), so the tail of one forward (ConvTWA steps, decoder) overlaps the head of the next.
    Per-request arithmetic and results are unchanged (bitwise: tests/test_hip_e2e.py); one request's latency grows,
    throughput rises -- 1890 -> 2068-2087 frames/s fp32 at one 8-frame clip per request, two streams
    (profiles/r5_experiments.md; round 2: 1505 -> 1607).
    Requests of the SAME video are ordered by their state: `predict_video` overlaps those up to the recurrence."""

    def __init__(self, model, streams: int = 2):
        self.model = model
        self.models = _inflight_replicas(model, max(1, int(streams)))      # lane-less handles, see there
        self._streams = None
        self._k = 0

    @torch.no_grad()
    def forward_clips(self, x, cb, state=None):
        """As `UAVSal.forward_clips`, asynchronous: returns (maps, state, event).  The outputs are produced on the
        replica's stream: wait for `event` (or call `synchronize()`) before using them on another stream."""
        dev = x.device
        if self._streams is None:
            self._streams = _host_streams(dev, len(self.models))
        i = self._k % len(self.models)
        if self._k % len(self.models) == 0:
            self.models = _inflight_replicas(self.model, len(self.models))      # follow the model's settings / weight edits
        self._k += 1
        s = self._streams[i]
        s.wait_stream(torch.cuda.current_stream(dev))        # inputs were produced on the caller's stream
        with torch.cuda.stream(s):
            for t in [x, *cb] + ([state] if torch.is_tensor(state) else []):
                t.record_stream(s)
            out, st = self.models[i].forward_clips(x, cb, state)
            ev = torch.cuda.Event()
            ev.record(s)
        return out, st, ev

    def synchronize(self):
        """Waits for every request in flight and raises if any of them reported a device error."""
        for s in self._streams or []:
            s.synchronize()
        for m in self.models:
            m.check_errors()

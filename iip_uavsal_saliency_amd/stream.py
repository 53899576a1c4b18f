"""Video-level driver: the loop of the reference's `test()` (Demo_Test.py:65-95) with frames,
recurrent state and outputs kept on the device.  The reference decodes a video with cv2,
normalises each group of `batch_size * time_dims` frames on the host, copies it to the GPU, runs
`model(x, cb, state)`, copies the maps back and resizes them one by one with cv2.  Here the
caller hands over the uint8 RGB frames as a tensor (decoding is out of scope: no video codec in
this image); normalisation happens inside the stem kernel, the state never leaves HBM, and the
maps are resized / normalised / quantised by `uavsal_postprocess`."""
from __future__ import annotations

import math
from typing import Optional

import torch

from . import ops


@torch.no_grad()
def predict_video(model, frames_u8: torch.Tensor, gauss_prior: torch.Tensor, ob_prior: torch.Tensor,
                  batch_size: int = 4, out_size: Optional[tuple] = None, return_maps: bool = False,
                  persistent_state: bool = True):
    """`frames_u8` uint8 `[F,3,H,W]` RGB (already letterboxed to the model size, as
    preprocess_videos does, utils_data.py:255-287), `gauss_prior` `[8,h,w]`, `ob_prior` `[20,h,w]`
    float32 (one map set, repeated per frame like get_bias, Demo_Test.py:14-27).
    Frames beyond the last full `time_dims` chunk are dropped (Demo_Test.py:68-70); groups of
    `batch_size * time_dims` frames are pushed through `model.forward` with the state carried
    (Demo_Test.py:75-86).  Returns uint8 `[F', H_out, W_out]` on the device (the reference's
    `pred_mat[..., 0]`), and the raw maps if asked."""
    dev = next(model.parameters()).device
    T = model.time_dims
    F = frames_u8.shape[0]
    count_bs = F // T
    keep = count_bs * T
    if keep < 2:
        raise RuntimeError("need at least one full chunk of time_dims >= 2 frames")
    frames_u8 = frames_u8[:keep].to(dev)
    H, W = frames_u8.shape[2:]
    out_size = out_size or (H, W)
    group = batch_size * T
    steps = math.ceil(count_bs / batch_size)
    state = None
    maps = []
    was = model.persistent_state
    model.persistent_state = bool(persistent_state)
    try:
        for i in range(steps):
            x = frames_u8[i * group:(i + 1) * group]
            n = x.shape[0]
            cb = [gauss_prior.to(dev).unsqueeze(0).expand(n, -1, -1, -1).contiguous(),
                  ob_prior.to(dev).unsqueeze(0).expand(n, -1, -1, -1).contiguous()]
            out, st = model(x, cb, state)
            # persistent mode: st[0] is a view of the engine's state buffer (valid until the next call, which
            # recognises it by address); a shorter last group runs on another plan, which loads it as a tensor
            state = [st[0].detach()]
            maps.append(out)
    finally:
        model.persistent_state = was
    maps = torch.cat(maps, 0)
    sal = ops.postprocess_predictions(maps, out_size[0], out_size[1])
    return (sal, maps) if return_maps else sal

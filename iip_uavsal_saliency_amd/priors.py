"""Centre-bias priors of the reference caller without cv2 / hdf5storage.

`get_bias` mirrors `Demo_Test.get_bias` (Demo_Test.py:14-27): it returns the two NCHW float32
tensors `[gauss [n,8,h,w], ob [n,20,h,w]]` the model's `cb` argument expects.
  * gaussian priors: `get_guasspriors` (utils_data.py:449-469) -- the closed form of
    `st_get_gaussmaps` + per-channel min-max, which is bit-identical to the shipped
    `gauss_priors.mat` (checked in tests/test_host_cpu.py), or the file itself if given;
  * observed priors: `get_ob_priors` / `read_ob_priors` (utils_data.py:552-604) read
    `<DATASET>_ob_priors_train.mat` (MATLAB v7.3 = HDF5) through `matio.loadmat`.
Resized priors: when the stored map size differs from the requested one the reference letterboxes
each map with `padding()` (utils_data.py:321-343) -- cv2.resize (INTER_LINEAR) to the largest size of the
same aspect ratio that fits, pasted centred into a zero array of dtype **uint8** (utils_data.py:460-464,
595-599) -- so the [0,1] floats are truncated to {0,1} (only exact 1.0 survives).  `quirk=True` (default)
reproduces that rule -- pinned to cv2's DOCUMENTED half-pixel INTER_LINEAR mapping, not to cv2 itself (absent here), and the
maps are resized in float32 where cv2 would resize a float64 prior file in double: a value that lands within one float32 ulp
of 1.0 can fall on the other side of the truncation; `quirk=False` keeps the resized floats.
Host-side numpy; this is caller code, not part of the device path.
"""
from __future__ import annotations

import os
from typing import List, Optional

import numpy as np
import torch

from . import matio, synth


def _load_maps(path: str) -> np.ndarray:
    if path.endswith(".npz"):
        return np.load(path)["PriorMaps"].astype(np.float32)
    return matio.loadmat(path)["PriorMaps"].astype(np.float32)


def resize_linear(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """cv2.resize(img, (out_w, out_h)) for a 2-D float32 map with the default INTER_LINEAR: half-pixel centres
    (`src = (dst + 0.5) * (src_size / dst_size) - 0.5`, computed in double, cast to float), edge samples replicated,
    horizontal pass then vertical pass in fp32."""
    h, w = img.shape
    img = np.asarray(img, dtype=np.float32)

    def taps(n_out, n_in):
        f = ((np.arange(n_out, dtype=np.float64) + 0.5) * (float(n_in) / float(n_out)) - 0.5).astype(np.float32)
        i0 = np.floor(f).astype(np.int64)
        f = (f - i0.astype(np.float32)).astype(np.float32)
        f[i0 < 0] = 0.0
        i0[i0 < 0] = 0
        f[i0 >= n_in - 1] = 0.0
        i0[i0 >= n_in - 1] = n_in - 1
        return i0, np.minimum(i0 + 1, n_in - 1), f

    x0, x1, fx = taps(out_w, w)
    y0, y1, fy = taps(out_h, h)
    rows = (img[:, x0] * (np.float32(1) - fx)[None, :]).astype(np.float32) + (img[:, x1] * fx[None, :]).astype(np.float32)
    out = (rows[y0] * (np.float32(1) - fy)[:, None]).astype(np.float32) + (rows[y1] * fy[:, None]).astype(np.float32)
    return out.astype(np.float32)


def letterbox(img: np.ndarray, shape_r: int, shape_c: int, quirk: bool = True) -> np.ndarray:
    """`padding(img, shape_r, shape_c, 1)` of the reference (utils_data.py:321-343) for one 2-D map: resize to the
    largest size of the source's aspect ratio that fits `shape_r x shape_c` (integer floor, as the reference computes
    it) and paste it centred into zeros.  `quirk=True`: the destination is uint8, as in the reference -- the float
    values are truncated toward zero on assignment; `quirk=False`: a float32 destination."""
    out = np.zeros((shape_r, shape_c), dtype=np.uint8 if quirk else np.float32)
    r0, c0 = img.shape
    if r0 / shape_r > c0 / shape_c:
        new_cols = min((c0 * shape_r) // r0, shape_c)
        rs = resize_linear(img, shape_r, (c0 * shape_r) // r0)[:, :new_cols]
        x = (shape_c - new_cols) // 2
        out[:, x:x + new_cols] = rs          # (uint8 destination: C cast, i.e. truncation)
    else:
        new_rows = min((r0 * shape_c) // c0, shape_r)
        rs = resize_linear(img, (r0 * shape_c) // c0, shape_c)[:new_rows]
        y = (shape_r - new_rows) // 2
        out[y:y + new_rows, :] = rs
    return out


def _fit(ims: np.ndarray, shape_r: int, shape_c: int, quirk: bool) -> np.ndarray:
    """utils_data.py:460-464 / 595-599: maps stored at another size are letterboxed channel by channel."""
    if ims.shape[0] == shape_r and ims.shape[1] == shape_c:
        return ims
    out = np.zeros((shape_r, shape_c, ims.shape[2]), dtype=np.uint8 if quirk else np.float32)
    for i in range(ims.shape[2]):
        out[:, :, i] = letterbox(ims[:, :, i], shape_r, shape_c, quirk)
    return out


def get_guasspriors(b_s: int = 2, shape_r: int = 45, shape_c: int = 80, channels: int = 8,
                    path: Optional[str] = None, quirk: bool = True) -> np.ndarray:
    """`[b_s, shape_r, shape_c, channels]` (utils_data.py:449-469): float32, or uint8 {0,1} when the file's maps had to
    be resized and `quirk` is set (the reference's behaviour).  Without a file: the closed form at the requested size
    (what the reference computes -- and saves -- when `gauss_priors.mat` does not exist yet)."""
    if path and os.path.exists(path):
        ims = _fit(_load_maps(path), shape_r, shape_c, quirk)
    else:
        ims = synth.gauss_priors(1, shape_r, shape_c, channels)[0].transpose(1, 2, 0)
    return np.repeat(ims[None], b_s, axis=0)


def get_ob_priors(path: str, b_s: int = 2, shape_r: int = 45, shape_c: int = 80, quirk: bool = True) -> np.ndarray:
    """`[b_s, shape_r, shape_c, 20]` from `<DATASET>_ob_priors_train.mat` (utils_data.py:591-604); resized maps as in
    `get_guasspriors`."""
    if not os.path.exists(path):
        raise ValueError("observed-prior file not found: %s" % path)
    ims = _fit(_load_maps(path), shape_r, shape_c, quirk)
    return np.repeat(ims[None], b_s, axis=0)


def get_bias(bias_type=(1, 1, 1), batch_size: int = 2, shape_r: int = 45, shape_c: int = 80,
             ob_prior_path: Optional[str] = None, gauss_prior_path: Optional[str] = None,
             device="cuda", quirk: bool = True, broadcast: bool = True) -> List[torch.Tensor]:
    """`[x_cb_gauss [n,8,h,w], x_cb_ob [n,20,h,w]]` float32 on `device` (Demo_Test.py:14-27).  `quirk`: see the module
    docstring (resized priors become {0,1} maps in the reference; False keeps the bilinear floats).
    `broadcast` (default): the n frames are a zero-stride view of ONE map set on the device -- the same values as the
    reference's `np.repeat` (utils_data.py:466-467, 601-602) without n copies, and `UAVSal.forward` recognises the view and
    runs its prior nets once (model.dedupe_priors); False materialises the n copies as the reference does."""
    def frames(maps_hwc):
        t = torch.from_numpy(np.ascontiguousarray(maps_hwc.transpose(2, 0, 1))).float().to(device)[None]
        return t.expand(batch_size, -1, -1, -1) if broadcast else t.repeat(batch_size, 1, 1, 1)
    if bias_type[0]:
        g = frames(get_guasspriors(1, shape_r, shape_c, 8, gauss_prior_path, quirk)[0])
    else:
        g = torch.tensor([]).float().to(device)
    if bias_type[1]:
        if ob_prior_path is None:
            raise ValueError("ob_prior_path (e.g. UAV2_ob_priors_train.mat) is required when bias_type[1] is set")
        o = frames(get_ob_priors(ob_prior_path, 1, shape_r, shape_c, quirk)[0])
    else:
        o = torch.tensor([]).float().to(device)
    return [g, o]

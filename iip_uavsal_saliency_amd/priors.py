"""Centre-bias priors of the reference caller without cv2 / hdf5storage.

`get_bias` mirrors `Demo_Test.get_bias` (Demo_Test.py:14-27): it returns the two NCHW float32
tensors `[gauss [n,8,h,w], ob [n,20,h,w]]` the model's `cb` argument expects.
  * gaussian priors: `get_guasspriors` (utils_data.py:449-469) -- the closed form of
    `st_get_gaussmaps` + per-channel min-max, which is bit-identical to the shipped
    `gauss_priors.mat` (checked in tests/test_host_cpu.py), or the file itself if given;
  * observed priors: `get_ob_priors` / `read_ob_priors` (utils_data.py:552-604) read
    `<DATASET>_ob_priors_train.mat` (MATLAB v7.3 = HDF5) through `matio.loadmat`.
Not reproduced: when the stored map size differs from the requested one the reference
letterboxes each map with cv2 into a *uint8* array (utils_data.py:460-464, 595-599), which
truncates the [0,1] floats to {0,1}; that path needs cv2.resize and raises here.
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


def _check_size(ims: np.ndarray, shape_r: int, shape_c: int, what: str) -> np.ndarray:
    if ims.shape[0] != shape_r or ims.shape[1] != shape_c:
        raise NotImplementedError(
            "%s are stored at %dx%d but %dx%d was requested: the reference resizes them with cv2 into a "
            "uint8 array (utils_data.py:460-464, 595-599); that quirk is not reproduced" % (
                what, ims.shape[0], ims.shape[1], shape_r, shape_c))
    return ims


def get_guasspriors(b_s: int = 2, shape_r: int = 45, shape_c: int = 80, channels: int = 8,
                    path: Optional[str] = None) -> np.ndarray:
    """float32 `[b_s, shape_r, shape_c, channels]` (utils_data.py:449-469)."""
    if path and os.path.exists(path):
        ims = _check_size(_load_maps(path), shape_r, shape_c, "gaussian priors")
    else:
        ims = synth.gauss_priors(1, shape_r, shape_c, channels)[0].transpose(1, 2, 0)
    return np.repeat(ims[None], b_s, axis=0)


def get_ob_priors(path: str, b_s: int = 2, shape_r: int = 45, shape_c: int = 80) -> np.ndarray:
    """float32 `[b_s, shape_r, shape_c, 20]` from `<DATASET>_ob_priors_train.mat` (utils_data.py:591-604)."""
    if not os.path.exists(path):
        raise ValueError("observed-prior file not found: %s" % path)
    ims = _check_size(_load_maps(path), shape_r, shape_c, "observed priors")
    return np.repeat(ims[None], b_s, axis=0)


def get_bias(bias_type=(1, 1, 1), batch_size: int = 2, shape_r: int = 45, shape_c: int = 80,
             ob_prior_path: Optional[str] = None, gauss_prior_path: Optional[str] = None,
             device="cuda") -> List[torch.Tensor]:
    """`[x_cb_gauss [n,8,h,w], x_cb_ob [n,20,h,w]]` on `device` (Demo_Test.py:14-27)."""
    if bias_type[0]:
        g = torch.from_numpy(get_guasspriors(batch_size, shape_r, shape_c, 8, gauss_prior_path)
                             .transpose(0, 3, 1, 2).copy()).float()
    else:
        g = torch.tensor([]).float()
    if bias_type[1]:
        if ob_prior_path is None:
            raise ValueError("ob_prior_path (e.g. UAV2_ob_priors_train.mat) is required when bias_type[1] is set")
        o = torch.from_numpy(get_ob_priors(ob_prior_path, batch_size, shape_r, shape_c)
                             .transpose(0, 3, 1, 2).copy()).float()
    else:
        o = torch.tensor([]).float()
    return [g.to(device), o.to(device)]

"""Deterministic synthetic weights, frames and priors for the UAVSal hot path.

There are no pretrained weights in the container (reference `weights/readme.md:1-7`
only links to OneDrive) and no datasets, so every test, golden vector and
benchmark uses closed-form data that is regenerated bit-identically on any box:
a counter-based integer hash (splitmix64 over `fnv1a(name) ^ seed` + flat index)
turned into uniforms with pure integer / float64 NumPy ops.  No
`torch.manual_seed` stream is involved, so values do not depend on the torch
build.

Tensor conventions follow the reference callers:
  * frames: uint8 RGB, then `/255` and ImageNet mean/std
    (reference `utils_data.py:43-65` `normalize_data`);
  * gaussian priors: reference `utils_data.py:391-412` `st_get_gaussmaps` followed
    by the per-channel min-max of `get_guasspriors` (`utils_data.py:453-456`);
  * observed priors: the reference reads `*_ob_priors_train.mat` (HDF5, no reader
    in this image) -> replaced by smooth seeded maps in [0, 1] of the same shape.

BatchNorm running statistics come from `synth_calib.json` (per-layer scalar
mean/var measured once with `oracle/calibrate_synth.py`, plus a hashed
per-channel jitter) so activations neither die nor saturate through the ~60
conv layers and the saliency map spans (0, 1) instead of sitting at 0.5
(SURVEY.md H6).
"""
from __future__ import annotations

import json
import os
from typing import Dict, Iterable, Optional, Tuple

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)
_CALIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "synth_calib.json")

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
EPS = 2.2204e-16  # reference utils_data.py EPS


def fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def hash_uniform(name: str, n: int, seed: int = 0, stream: int = 0) -> np.ndarray:
    """`n` float64 uniforms in [0, 1) that depend only on (name, seed, stream, index)."""
    base = (fnv1a64(name) ^ ((seed * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)
            ^ ((stream * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF)) & 0xFFFFFFFFFFFFFFFF
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) * np.uint64(0x2545F4914F6CDD1D) + np.uint64(base)
    z = _splitmix64(_splitmix64(idx))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def hash_normal(name: str, n: int, seed: int = 0, stream: int = 0) -> np.ndarray:
    """Approximate N(0,1): sum of 4 uniforms, centred and rescaled (bounded, smooth)."""
    u = sum(hash_uniform(name, n, seed, stream * 4 + k) for k in range(4))
    return (u - 2.0) * np.sqrt(3.0)


# --------------------------------------------------------------------------- weights

def load_calibration() -> Dict[str, Tuple[float, float]]:
    if os.path.exists(_CALIB_PATH):
        with open(_CALIB_PATH) as f:
            raw = json.load(f)
        return {k: (float(v[0]), float(v[1])) for k, v in raw.get("bn", {}).items()}
    return {}


def bn_channel_stats(prefix: str, c: int, mean_s: float, var_s: float, seed: int = 0):
    """Per-channel running_mean / running_var from the per-layer scalars + hashed jitter."""
    n1 = hash_normal(prefix + ".running_mean", c, seed)
    u2 = hash_uniform(prefix + ".running_var", c, seed)
    mean = mean_s + 0.25 * np.sqrt(max(var_s, 1e-12)) * n1
    var = max(var_s, 1e-12) * (0.6 + 0.8 * u2)
    return mean.astype(np.float32), var.astype(np.float32)


def synth_tensor(key: str, shape: Iterable[int], seed: int = 0,
                 calib: Optional[Dict[str, Tuple[float, float]]] = None) -> np.ndarray:
    """Value of state_dict entry `key` (reference schema)."""
    shape = tuple(int(s) for s in shape)
    n = int(np.prod(shape)) if len(shape) else 1
    leaf = key.rsplit(".", 1)[-1]
    prefix = key.rsplit(".", 1)[0]
    if leaf == "num_batches_tracked":
        return np.zeros(shape, dtype=np.int64)
    if len(shape) == 4:  # conv weight [Cout, Cin/groups, kh, kw]
        fan_in = shape[1] * shape[2] * shape[3]
        std = np.sqrt(2.0 / fan_in)
        w = (hash_uniform(key, n, seed) * 2.0 - 1.0) * (np.sqrt(3.0) * std)
        return w.reshape(shape).astype(np.float32)
    if len(shape) == 1:
        c = shape[0]
        if leaf == "weight":     # BN gamma
            g = 0.8 + 0.4 * hash_uniform(key, c, seed)
            if key == "conv_out_st.conv.3.weight":
                g = np.full((c,), 2.5)   # logits std ~2.5 -> map spans (0,1)
            return g.astype(np.float32)
        if leaf == "bias":       # BN beta
            b = 0.6 * hash_uniform(key, c, seed) - 0.1
            if key == "conv_out_st.conv.3.bias":
                b = np.full((c,), -0.2)
            return b.astype(np.float32)
        if leaf in ("running_mean", "running_var"):
            m_s, v_s = (calib or {}).get(prefix, (0.0, 1.0))
            mean, var = bn_channel_stats(prefix, c, m_s, v_s, seed)
            return mean if leaf == "running_mean" else var
    raise ValueError(f"synth_tensor: unexpected entry {key} {shape}")


def synth_state_dict(shapes: Dict[str, Tuple[int, ...]], seed: int = 0,
                     calib: Optional[Dict[str, Tuple[float, float]]] = None):
    """`{key: torch.Tensor}` for every `(key, shape)`; load with `load_state_dict`."""
    import torch
    if calib is None:
        calib = load_calibration()
    out = {}
    for k, shp in shapes.items():
        out[k] = torch.from_numpy(synth_tensor(k, shp, seed, calib))
    return out


def load_synth_weights(model, seed: int = 0, calib=None):
    """Fill `model` (any module with the reference state_dict schema) in place."""
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = synth_state_dict(shapes, seed, calib)
    model.load_state_dict(sd, strict=True)
    return model


# --------------------------------------------------------------------------- inputs

def synth_frames_u8(n_frames: int, height: int, width: int, seed: int = 0, t0: int = 0) -> np.ndarray:
    """uint8 RGB frames `[n, 3, H, W]`: three moving gaussian blobs over drifting
    sinusoids plus a little hashed texture, so consecutive frames differ smoothly
    (exercises the temporal differences of `teConv_sub`, reference model.py:194-198)."""
    ys = (np.arange(height, dtype=np.float64) / max(height - 1, 1))[:, None]
    xs = (np.arange(width, dtype=np.float64) / max(width - 1, 1))[None, :]
    par = hash_uniform("frames.params", 64, seed)
    tex = hash_uniform("frames.texture", 3 * height * width, seed).reshape(3, height, width)
    out = np.empty((n_frames, 3, height, width), dtype=np.uint8)
    for i in range(n_frames):
        t = float(t0 + i)
        img = np.zeros((3, height, width), dtype=np.float64)
        for c in range(3):
            fx, fy = 2.0 + 5.0 * par[c], 1.5 + 4.0 * par[3 + c]
            ph = 6.28318 * par[6 + c] + 0.11 * t * (1 + c)
            img[c] = 0.35 + 0.18 * np.sin(6.28318 * (fx * xs + fy * ys) + ph)
        for b in range(3):
            cx = (par[10 + b] + 0.013 * (1 + b) * t) % 1.0
            cy = (par[14 + b] + 0.009 * (2 - b) * t + 0.05 * np.sin(0.3 * t + b)) % 1.0
            sg = 0.04 + 0.08 * par[18 + b]
            blob = np.exp(-((xs - cx) ** 2 + (ys - cy) ** 2) / (2 * sg * sg))
            for c in range(3):
                img[c] += (0.25 + 0.5 * par[22 + 3 * b + c]) * blob
        img += 0.06 * (tex - 0.5)
        out[i] = np.clip(np.rint(img * 255.0), 0, 255).astype(np.uint8)
    return out


def normalize_frames(frames_u8: np.ndarray) -> np.ndarray:
    """uint8 `[n,3,H,W]` -> float32 ImageNet-normalised (reference utils_data.py:43-65)."""
    ims = frames_u8.astype(np.float32) / 255.0
    for c in range(3):
        ims[:, c] = (ims[:, c] - IMAGENET_MEAN[c]) / IMAGENET_STD[c]
    return ims


def gauss_priors(n: int, h: int, w: int, channels: int = 8) -> np.ndarray:
    """float32 `[n, channels, h, w]`: closed form of reference `st_get_gaussmaps`
    (utils_data.py:391-412) + per-channel min-max (utils_data.py:453-456), NCHW as
    produced by `Demo_Test.get_bias` (Demo_Test.py:14-18)."""
    e = h / w
    e1 = (1 - e) / 2
    e2 = e1 + e
    sigma = e * np.arange(1, channels + 1, dtype=np.float64) / 16
    x_t = np.ones((h, 1)) @ np.linspace(0.0, 1.0, w).reshape(1, w)
    y_t = np.linspace(e1, e2, h).reshape(h, 1) @ np.ones((1, w))
    x_t = np.repeat(x_t[:, :, None], channels, axis=2)
    y_t = np.repeat(y_t[:, :, None], channels, axis=2)
    g = 1 / (2 * np.pi * sigma * sigma + EPS) * np.exp(
        -((x_t - 0.5) ** 2 / (2 * sigma ** 2 + EPS) + (y_t - 0.5) ** 2 / (2 * sigma ** 2 + EPS)))
    g = (g - g.min((0, 1))) / (g.max((0, 1)) - g.min((0, 1)) + EPS)
    g = g.astype(np.float32).transpose(2, 0, 1)[None]
    return np.ascontiguousarray(np.repeat(g, n, axis=0))


def ob_priors(n: int, h: int, w: int, channels: int = 20, seed: int = 0) -> np.ndarray:
    """float32 `[n, channels, h, w]` smooth maps in [0,1] standing in for the
    observed-fixation priors of reference `get_ob_priors` (utils_data.py:591-604)."""
    ys = (np.arange(h, dtype=np.float64) / max(h - 1, 1))[:, None]
    xs = (np.arange(w, dtype=np.float64) / max(w - 1, 1))[None, :]
    par = hash_uniform("ob_priors.params", channels * 8, seed).reshape(channels, 8)
    maps = np.empty((channels, h, w), dtype=np.float64)
    for c in range(channels):
        p = par[c]
        m = np.exp(-((xs - p[0]) ** 2 / (2 * (0.08 + 0.3 * p[2]) ** 2)
                     + (ys - p[1]) ** 2 / (2 * (0.08 + 0.3 * p[3]) ** 2)))
        m += 0.5 * p[4] * np.exp(-((xs - p[5]) ** 2 + (ys - p[6]) ** 2) / (2 * 0.1 ** 2))
        m += 0.1 * p[7]
        maps[c] = m / m.max()
    maps = maps.astype(np.float32)[None]
    return np.ascontiguousarray(np.repeat(maps, n, axis=0))

"""ORACLE (test infrastructure): numpy restatement of the reference caller's output
post-processing -- `postprocess_predictions` (reference utils_data.py:289-303) followed by
`np2mat`/`im2uint8` (utils_data.py:68-82), as used at Demo_Test.py:89-91.

Pin: OpenCV is not installed in this container and the reference holds no fixture for this step, so the resize
(`cv2.resize`, default INTER_LINEAR) is restated from OpenCV's documented rule for float images (half-pixel centres,
`src = (dst + 0.5) * (src_size / dst_size) - 0.5`, coordinates in double cast to float, border replicated, horizontal
pass then vertical pass in fp32).  It is pinned by known answers that follow from that rule and from the crop
arithmetic of utils_data.py:289-303 alone (tests/post_vectors.py: linear ramps incl. clamped borders, a one-hot map's
four weights, both crop branches), NOT by outputs of cv2 itself: a deviation of cv2's fixed-point coefficient tables
from the documented rule would not be seen.
"""
import numpy as np


def _resize_linear(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    h, w = img.shape
    img = img.astype(np.float32)

    def coeffs(n_out, n_in):
        scale = float(n_in) / float(n_out)
        f = ((np.arange(n_out, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(np.float32)).astype(np.float32)
        lo = s < 0
        s[lo] = 0
        f[lo] = 0.0
        hi = s >= n_in - 1
        s[hi] = n_in - 1
        f[hi] = 0.0
        s1 = np.minimum(s + 1, n_in - 1)
        return s, s1, f

    sx, sx1, fx = coeffs(out_w, w)
    sy, sy1, fy = coeffs(out_h, h)
    a0, a1 = (np.float32(1.0) - fx).astype(np.float32), fx
    rows = (img[:, sx] * a0[None, :]).astype(np.float32) + (img[:, sx1] * a1[None, :]).astype(np.float32)
    b0, b1 = (np.float32(1.0) - fy).astype(np.float32), fy
    out = (rows[sy] * b0[:, None]).astype(np.float32) + (rows[sy1] * b1[:, None]).astype(np.float32)
    return out.astype(np.float32)


def postprocess_predictions(pred: np.ndarray, shape_r: int, shape_c: int) -> np.ndarray:
    """utils_data.py:289-303 for one `[h, w]` map -> float32 `[shape_r, shape_c]` in [0, 255]."""
    h, w = pred.shape
    rows_rate = shape_r / h
    cols_rate = shape_c / w
    if rows_rate > cols_rate:
        new_cols = (w * shape_r) // h
        r = _resize_linear(pred, shape_r, new_cols)
        x0 = (r.shape[1] - shape_c) // 2
        img = r[:, x0:x0 + shape_c]
    else:
        new_rows = (h * shape_c) // w
        r = _resize_linear(pred, new_rows, shape_c)
        y0 = (r.shape[0] - shape_r) // 2
        img = r[y0:y0 + shape_r, :]
    return (img / np.max(img) * np.float32(255)).astype(np.float32)


def to_uint8(img: np.ndarray) -> np.ndarray:
    """utils_data.py:68-82 (np2mat / im2uint8)."""
    img = np.clip(img, 0, 255)
    return np.rint(img).astype(np.uint8)

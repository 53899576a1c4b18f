"""Known-answer vectors for the caller's output post-processing (reference utils_data.py:289-303 +
np2mat/im2uint8 :68-82) that need neither cv2 nor the restatement: they follow from OpenCV's documented
INTER_LINEAR rule alone -- destination pixel X samples the source at `sx = (X + 0.5) * (w / W) - 0.5`, clamped to
[0, w - 1] (edge replication), and interpolates linearly between floor(sx) and floor(sx) + 1.
  * a linear ramp `a*x + b*y + c` is reproduced exactly by linear interpolation, so the resized map is the ramp
    evaluated at the clamped sample positions;
  * a one-hot map upscaled by exactly 2 spreads into the outer product of [0.25, 0.75, 0.75, 0.25];
  * the centre crop of `postprocess_predictions` is then an index shift: (new - shape) // 2.
Each case: (pred [h,w] float32, shape_r, shape_c, expected float64 [shape_r, shape_c] BEFORE rint, in [0, 255])."""
import numpy as np


def _sample_pos(n_out, n_in):
    return np.clip((np.arange(n_out, dtype=np.float64) + 0.5) * (n_in / n_out) - 0.5, 0.0, n_in - 1.0)


def _crop_geometry(h, w, shape_r, shape_c):
    """utils_data.py:289-300: (resized rows, resized cols, first row, first col)."""
    if shape_r / h > shape_c / w:
        new_cols = (w * shape_r) // h
        return shape_r, new_cols, 0, (new_cols - shape_c) // 2
    new_rows = (h * shape_c) // w
    return new_rows, shape_c, (new_rows - shape_r) // 2, 0


def ramp_case(h, w, shape_r, shape_c, a=0.004, b=0.007, c=0.05):
    ys, xs = np.mgrid[0:h, 0:w]
    pred = (a * xs + b * ys + c).astype(np.float32)
    R, Cc, y0, x0 = _crop_geometry(h, w, shape_r, shape_c)
    sy, sx = _sample_pos(R, h), _sample_pos(Cc, w)
    full = a * sx[None, :] + b * sy[:, None] + c
    img = full[y0:y0 + shape_r, x0:x0 + shape_c]
    return pred, shape_r, shape_c, img / img.max() * 255.0


def one_hot_case(h=9, w=13, y=4, x=6):
    pred = np.zeros((h, w), np.float32)
    pred[y, x] = 0.8
    k = np.array([0.25, 0.75, 0.75, 0.25])
    exp = np.zeros((2 * h, 2 * w))
    exp[2 * y - 1:2 * y + 3, 2 * x - 1:2 * x + 3] = np.outer(k, k) * 0.8
    return pred, 2 * h, 2 * w, exp / exp.max() * 255.0


def cases():
    return [
        ("ramp, same aspect 45x80 -> 360x640", ) + ramp_case(45, 80, 360, 640),
        ("ramp, rows_rate > cols_rate: 45x80 -> 360x600, resized to 640 columns, columns 20..619 kept", ) + ramp_case(45, 80, 360, 600),
        ("ramp, rows_rate < cols_rate: 45x80 -> 300x640, resized to 360 rows, rows 30..329 kept", ) + ramp_case(45, 80, 300, 640),
        ("ramp, 36x64 -> 288x512", ) + ramp_case(36, 64, 288, 512, a=0.009, b=-0.003, c=0.4),
        ("one-hot x2", ) + one_hot_case(),
    ]

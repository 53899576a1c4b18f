"""Host-side weight preparation for the HIP kernels: BatchNorm folding and the packed
weight layouts documented in include/uavsal_hip.h.  Pure tensor reshuffling on the
host, done once per engine build (never in the timed path)."""
from __future__ import annotations

from typing import Optional, Tuple

import torch

BN_EPS = 1e-5  # nn.BatchNorm2d default; the reference never overrides it (model.py:70,95)

# order of the 32 k's of one K tile in the bf16 layouts: chunk c = {4c..4c+3, 16+4c..16+4c+3}
_K_PERM32 = [k for c in range(4) for k in (list(range(4 * c, 4 * c + 4)) + list(range(16 + 4 * c, 16 + 4 * c + 4)))]


def fold_bn(bn: torch.nn.BatchNorm2d) -> Tuple[torch.Tensor, torch.Tensor]:
    """Eval-mode BatchNorm as y = x*scale + bias (fp64 fold, fp32 result)."""
    g = bn.weight.detach().double().cpu()
    b = bn.bias.detach().double().cpu()
    m = bn.running_mean.detach().double().cpu()
    v = bn.running_var.detach().double().cpu()
    scale = g / torch.sqrt(v + bn.eps)
    bias = b - m * scale
    return scale.float(), bias.float()


def pad_vec(v: torch.Tensor, n: int, fill: float = 0.0) -> torch.Tensor:
    out = torch.full((n,), fill, dtype=torch.float32)
    out[: v.numel()] = v.float().cpu()
    return out


def roundup(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def pack_conv_weight(w: torch.Tensor, prec: str) -> torch.Tensor:
    """`w` [Cout, Cin, kh, kw] (kh=kw in {1,3}) -> packed byte tensor (uint8, 1-D) in the layout
    `uavsal_conv_gemm` expects for `prec` in {'f32','bf16','bf16x3','f16x3'}, or 'f32k32': fp32 for the kernels
    with 32-float K stages (uavsal_conv_tile 8 / 9: the 3x3 K order then has 32-channel blocks), or 'f16x3i': the f16x3
    values with the natural k order, hi and lo of one (K step, output channel) interleaved into one 128-byte
    line [Kpad/32][Npad][hi 32 | lo 32] (the pre-split LDS-DMA path, uavsal_conv_uses_split), or 'f16x3j' (below)."""
    w = w.detach().float().cpu()
    cout, cin, kh, kw = w.shape
    taps = kh * kw
    assert taps in (1, 9) and kh == kw
    kt = 16 if prec == "f32" else 32            # ('f16x3j' pads K to 32 too: the kernel walks Cin / 16 steps of it)
    if prec == "f32k32":
        prec = "f32"
    if taps == 9 and cin % 32:
        raise RuntimeError("3x3 dense conv needs Cin % 32 == 0")
    k = taps * cin
    kpad, npad = roundup(k, kt), roundup(cout, 32)
    m = torch.zeros(npad, kpad, dtype=torch.float32)
    if taps == 1:
        m[:cout, :k] = w.reshape(cout, k)
    else:
        # 3x3: channel-block-major, tap-minor: k = ((ci // kt) * 9 + tap) * kt + ci % kt, so the nine taps
        # of one channel chunk are nine consecutive K steps (activation lines are reused while in L2)
        if cin % kt:
            raise RuntimeError("3x3 dense conv needs Cin %% %d == 0" % kt)
        m[:cout, :k] = w.reshape(cout, cin // kt, kt, 9).permute(0, 1, 3, 2).reshape(cout, k)
    if prec == "f32":
        return m.contiguous().view(torch.uint8).reshape(-1)
    if prec == "f16x3j":    # LDS-halo depthwise -> projection kernel (uavsal_conv_dwproj): K steps of 16 channels,
        # natural k order, one 64-byte line [hi 16 | lo 16] per (K step, output channel): [Kpad/16][Npad][2][16]
        assert taps == 1
        mm = m.view(npad, kpad // 16, 16) * 64.0
        hi = mm.to(torch.float16)
        lo = (mm - hi.float()).to(torch.float16)
        return torch.stack([hi, lo], 2).permute(1, 0, 2, 3).contiguous().view(torch.uint8).reshape(-1)
    # 16-bit layouts are K-step-major: [Kpad/32][panel][Npad][32] -- one K step is one contiguous run
    idx = torch.tensor(list(range(32)) if prec == "f16x3i" else _K_PERM32, dtype=torch.long)
    m = m.view(npad, kpad // 32, 32)[:, :, idx]                 # [npad, steps, 32]
    if prec in ("f16x3", "f16x3i"):
        m = m * 64.0
        hi = m.to(torch.float16)
        lo = (m - hi.float()).to(torch.float16)
        if prec == "f16x3i":
            return torch.stack([hi, lo], 2).permute(1, 0, 2, 3).contiguous().view(torch.uint8).reshape(-1)
        return torch.stack([hi, lo], 0).permute(2, 0, 1, 3).contiguous().view(torch.uint8).reshape(-1)
    hi = m.to(torch.bfloat16)
    if prec == "bf16":
        return hi.permute(1, 0, 2).contiguous().view(torch.uint8).reshape(-1)
    lo = (m - hi.float()).to(torch.bfloat16)
    return torch.stack([hi, lo], 0).permute(2, 0, 1, 3).contiguous().view(torch.uint8).reshape(-1)


_WINO_G = {2: [[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]],
           4: [[1 / 4, 0.0, 0.0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6],
               [1 / 24, -1 / 12, 1 / 6], [0.0, 0.0, 1.0]]}


def pack_wino_weight(w: torch.Tensor, r: int = 2) -> torch.Tensor:
    """Dense 3x3 `w` [Cout, Cin, 3, 3] -> the P*P (P = r + 2) Winograd F(r x r, 3x3) filter matrices U_k = (G g G^T)_k,
    k = P i + j, each packed like an fp32 1x1 weight [Npad][Kpad] (Kpad = Cin rounded up to 32), one after another:
    fp32 [P*P][Npad][Kpad] as a flat byte tensor (uavsal_conv_desc.w_group_stride = Npad * Kpad).  Transformed in
    fp64, rounded once."""
    w = w.detach().double().cpu()
    cout, cin, kh, kw = w.shape
    assert kh == 3 and kw == 3 and r in _WINO_G
    g = torch.tensor(_WINO_G[r], dtype=torch.float64)
    pp = (r + 2) * (r + 2)
    u = torch.einsum("ij,ocjk,lk->iloc", g, w, g).reshape(pp, cout, cin)          # [k][o][c]
    kpad, npad = roundup(cin, 32), roundup(cout, 32)
    m = torch.zeros(pp, npad, kpad, dtype=torch.float32)
    m[:, :cout, :cin] = u.float()
    return m.contiguous().view(torch.uint8).reshape(-1)


def pack_dw_weight(w: torch.Tensor) -> torch.Tensor:
    """depthwise `w` [C, 1, 3, 3] -> tap-major [9, C] fp32."""
    c = w.shape[0]
    return w.detach().float().cpu().reshape(c, 9).t().contiguous()


def pack_stem_weight(w: torch.Tensor) -> torch.Tensor:
    """stem `w` [32, 3, 3, 3] -> [27, 32] fp32, row = ci*9 + ky*3 + kx."""
    return w.detach().float().cpu().reshape(32, 27).t().contiguous()

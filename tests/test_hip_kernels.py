"""Per-kernel parity on the MI355X: every entry point of the C ABI against a plain
PyTorch-CPU fp32 restatement of the same reference operator (F.conv2d / F.interpolate /
slicing), on seeded inputs.  Tolerances are written next to each check:
  f32    : exact-fp32 MFMA, only summation order differs      -> 2e-5 * scale
  f16x3  : split-fp16 (hi*hi + hi*lo + lo*hi), ~2^-21 relative -> 2e-5 * scale
  bf16x3 : split-bf16 (hi*hi + hi*lo + lo*hi), ~2^-16 relative -> 2e-4 * scale
  bf16   : single bf16 MFMA, ~2^-8 relative                    -> 3e-2 * scale
"""
import numpy as np
import ctypes

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = {"f32": 2e-5, "f16x3": 2e-5, "bf16x3": 2e-4, "bf16": 3e-2}


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from iip_uavsal_saliency_amd import ops as o
    return o


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def nchw(t):
    return t.permute(0, 3, 1, 2).contiguous().cpu()


def act_ref(y, act):
    if act == 1:
        return torch.clamp(y, 0, 6)
    if act == 2:
        return torch.sigmoid(y)
    return y


CONV1_CASES = [
    # n, h, w, cin, cout, act, res
    (2, 5, 7, 8, 48, 1, False),        # gauss prior expand (Cin=8 < K tile)
    (2, 5, 7, 20, 120, 1, False),      # observed prior expand (Cin=20, ragged K tile)
    (1, 9, 13, 32, 16, 0, False),      # features.1 pw-linear, Cout < 32
    (2, 12, 20, 96, 24, 0, False),
    (1, 12, 20, 144, 24, 0, True),     # residual
    (2, 12, 20, 256, 1536, 1, False),  # big N
    (1, 23, 40, 1536, 256, 0, True),   # big K
    (2, 12, 20, 1536, 1, 2, False),    # decoder: Cout=1 + sigmoid
    (3, 7, 5, 320, 256, 1, False),
    (1, 45, 80, 32, 256, 1, True),     # te last_conv + x_sp
    (2, 45, 80, 1536, 32, 1, False),   # 128x32 tile, long K: every wave must count the same DMA requests
    (2, 45, 80, 1536, 64, 0, True),    # fucb project
]


@pytest.mark.parametrize("prec", ["f32", "f16x3", "bf16x3", "bf16"])
@pytest.mark.parametrize("case", CONV1_CASES)
def test_conv1x1(ops, prec, case):
    n, h, w, cin, cout, act, use_res = case
    x = rnd((n, cin, h, w), 1, 2.0)
    wt = rnd((cout, cin, 1, 1), 2, 1.0 / np.sqrt(cin))
    scale = rnd((cout,), 3) * 0.5 + 1.0
    bias = rnd((cout,), 4)
    res = rnd((n, cout, h, w), 5) if use_res else None
    ref = act_ref(F.conv2d(x, wt) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1), act)
    if use_res:
        ref = ref + res
    got = ops.conv_gemm(nhwc(x), wt, scale, bias, act=act, res=nhwc(res) if use_res else None, prec=prec)
    err = (nchw(got) - ref).abs().max().item()
    assert err <= TOL[prec] * 4.0, (case, prec, err)


@pytest.mark.parametrize("tile", [1, 2, 3, 4, 5, 6, 7])
@pytest.mark.parametrize("prec", ["f32", "f16x3", "bf16x3"])
def test_conv1x1_all_tiles(ops, prec, tile):
    n, h, w, cin, cout = 2, 11, 13, 64, 160      # M = 286: ragged against every tile height
    x = rnd((n, cin, h, w), 11, 2.0)
    wt = rnd((cout, cin, 1, 1), 12, 1.0 / 8)
    ref = F.conv2d(x, wt)
    got = ops.conv_gemm(nhwc(x), wt, None, None, prec=prec, tile=tile)
    err = (nchw(got) - ref).abs().max().item()
    assert err <= TOL[prec] * 4.0, (tile, prec, err)


def test_conv_f32_256x128_tile(ops):
    """The 8-wave 256 x 128 instance of the fp32 LDS-DMA GEMM (tile 7): expand / project / 3x3 shapes, ragged M and
    Cout, BN + ReLU6 + residual."""
    for (n, h, w, cin, cout, taps, use_res) in [(3, 20, 23, 256, 1536, 1, False), (2, 20, 23, 1536, 256, 1, True),
                                               (2, 12, 15, 64, 256, 9, False), (2, 20, 23, 72, 200, 1, True),
                                               (1, 45, 80, 448, 256, 9, False)]:
        x = rnd((n, cin, h, w), 71, 2.0)
        k = 3 if taps == 9 else 1
        wt = rnd((cout, cin, k, k), 72, 1.0 / (cin * taps) ** 0.5)
        scale, bias = rnd((cout,), 73, 0.5) + 1.0, rnd((cout,), 74, 0.1)
        ref = torch.clamp(F.conv2d(x, wt, padding=k // 2) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1), 0, 6)
        res = rnd((n, cout, h, w), 75) if use_res else None
        if use_res:
            ref = ref + res
        got = ops.conv_gemm(nhwc(x), wt, scale, bias, act=1, res=nhwc(res) if use_res else None, prec="f32", tile=7)
        err = (nchw(got) - ref).abs().max().item()
        assert err <= TOL["f32"] * 4.0, ((n, h, w, cin, cout, taps), err)


@pytest.mark.parametrize("tile", [1, 5, 6])
@pytest.mark.parametrize("prec", ["f16x3", "bf16x3"])
@pytest.mark.parametrize("case", [(3, 20, 23, 96, 512, 1, True), (2, 20, 23, 1536, 256, 1, False),
                                  (2, 12, 15, 64, 256, 9, False), (2, 20, 23, 72, 200, 1, True)])
def test_conv_wide_tile(ops, prec, case, tile):
    """The big tiles of the split 16-bit precisions (128 x 128; 128 x 256 and 256 x 256 on 8 waves) on expand /
    project / 3x3 / ragged shapes, odd K-step counts, BN + ReLU6 + residual epilogue."""
    n, h, w, cin, cout, taps, use_res = case
    x = rnd((n, cin, h, w), 71, 2.0)
    k = 3 if taps == 9 else 1
    wt = rnd((cout, cin, k, k), 72, 1.0 / (cin * taps) ** 0.5)
    scale, bias = rnd((cout,), 73, 0.5) + 1.0, rnd((cout,), 74, 0.1)
    ref = F.conv2d(x, wt, padding=k // 2) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)
    ref = ref.clamp(0, 6)
    res = rnd((n, cout, h, w), 75, 1.0)
    if use_res:
        ref = ref + res
    got = ops.conv_gemm(nhwc(x), wt, scale, bias, act=1, res=nhwc(res) if use_res else None, prec=prec, tile=tile)
    err = (nchw(got) - ref).abs().max().item()
    assert err <= TOL[prec] * 4.0, (case, prec, err)


@pytest.mark.parametrize("case", [(8, 45, 80, 1536, 256, 1, True),     # 450 tiles, 96 K stages: the projections
                                  (8, 45, 80, 448, 256, 9, False),     # 450 tiles, 3x3 (conv_last)
                                  (8, 45, 80, 1024, 256, 1, True),     # 450 tiles, 64 K stages (the shortest taken)
                                  (8, 45, 80, 1040, 768, 1, False),    # 1350 tiles: ranges of 2.6 tiles
                                  (5, 45, 80, 1024, 256, 1, False),    # 282 tiles: ranges of 0.55 tile (3 pieces)
                                  (8, 45, 80, 256, 256, 1, True)])     # short K: stays whole-tile
def test_conv_stream_k(ops, case):
    """fp32 128x128 GEMM with the stream-K workspace: tiles whose K loop is split across workgroups must
    give the whole-tile result (up to fp32 summation order) and leave the workspace zeroed."""
    n, h, w, cin, cout, taps, use_res = case
    x = rnd((n, cin, h, w), 81, 2.0)
    k = 3 if taps == 9 else 1
    wt = rnd((cout, cin, k, k), 82, 1.0 / (cin * taps) ** 0.5)
    scale, bias = rnd((cout,), 83, 0.5) + 1.0, rnd((cout,), 84, 0.1)
    res = rnd((n, cout, h, w), 85, 1.0)
    xd, rd = nhwc(x), nhwc(res) if use_res else None
    whole = ops.conv_gemm(xd, wt, scale, bias, act=1, res=rd, prec="f32", tile=1)
    split = ops.conv_gemm(xd, wt, scale, bias, act=1, res=rd, prec="f32", tile=1, stream_k=True)
    assert (split - whole).abs().max().item() <= TOL["f32"] * 4.0
    ref = (F.conv2d(x, wt, padding=k // 2) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)).clamp(0, 6)
    if use_res:
        ref = ref + res
    assert (nchw(split) - ref).abs().max().item() <= TOL["f32"] * 4.0


def test_conv1x1_channel_slices(ops):
    """Reading from and writing into channel slices of wider buffers (how torch.cat disappears)."""
    n, h, w = 2, 6, 9
    wide_in = nhwc(rnd((n, 96, h, w), 21))
    wt = rnd((64, 32, 1, 1), 22, 0.2)
    wide_out = torch.full((n, h, w, 192), -7.0, device="cuda")
    ops.conv_gemm(wide_in[..., 32:64], wt, None, None, out=wide_out[..., 64:128])
    ref = F.conv2d(nchw(wide_in)[:, 32:64], wt)
    assert (nchw(wide_out)[:, 64:128] - ref).abs().max().item() <= 1e-4
    assert torch.all(wide_out[..., :64] == -7.0) and torch.all(wide_out[..., 128:] == -7.0)


CONV3_CASES = [(1, 9, 13, 32, 64), (2, 12, 20, 448, 256), (1, 45, 80, 256, 256), (3, 5, 4, 64, 32)]


@pytest.mark.parametrize("prec", ["f32", "f16x3", "bf16x3", "bf16"])
@pytest.mark.parametrize("case", CONV3_CASES)
def test_conv3x3(ops, prec, case):
    n, h, w, cin, cout = case
    x = rnd((n, cin, h, w), 31, 2.0)
    wt = rnd((cout, cin, 3, 3), 32, 1.0 / np.sqrt(9 * cin))
    scale = rnd((cout,), 33) * 0.5 + 1.0
    bias = rnd((cout,), 34)
    ref = torch.clamp(F.conv2d(x, wt, padding=1) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1), 0, 6)
    got = ops.conv_gemm(nhwc(x), wt, scale, bias, act=1, prec=prec)
    err = (nchw(got) - ref).abs().max().item()
    assert err <= TOL[prec] * 4.0, (case, prec, err)


WINO_CASES = [
    # n, h, w, cin, cout, act, res
    (1, 12, 20, 64, 96, 1, False), (2, 9, 13, 448, 256, 1, False),      # odd sizes: half-empty last tile row / column
    (1, 45, 80, 256, 256, 0, True), (3, 7, 5, 32, 40, 0, False), (2, 2, 3, 64, 64, 1, True), (8, 23, 40, 96, 128, 1, False)]


@pytest.mark.parametrize("case", [(8, 12, 20, 1920, 256, 3, True), (2, 9, 13, 96, 64, 2, False), (3, 12, 20, 512, 128, 4, True)])
def test_conv1x1_output_groups_with_their_own_inputs(ops, case):
    """uavsal_conv_desc.n_group: the three dilated ASPP projections (model.py:142-147) as ONE launch -- group g's output
    channels come from group g's input columns; with the workspace the K loop is shared out over workgroups."""
    n, h, w, cin, ng, groups, ws = case
    x = rnd((n, groups * cin, h, w), 301, 2.0)
    wt = rnd((groups * ng, cin, 1, 1), 302, 1.0 / np.sqrt(cin))
    scale = rnd((groups * ng,), 303) * 0.5 + 1.0
    bias = rnd((groups * ng,), 304)
    ref = torch.cat([F.conv2d(x[:, g * cin:(g + 1) * cin], wt[g * ng:(g + 1) * ng]) for g in range(groups)], 1)
    ref = ref * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)
    got = ops.conv_gemm(nhwc(x), wt, scale, bias, prec="f32", n_group=ng, stream_k=ws)
    err = (nchw(got) - ref).abs().max().item()
    assert err <= 3e-5 * max(1.0, ref.abs().max().item()), (case, err)
    with pytest.raises(RuntimeError):          # groups must be whole 64-column tiles
        ops.conv_gemm(nhwc(x), wt, scale, bias, prec="f32", n_group=ng // 2 + 8)


@pytest.mark.parametrize("r", [2, 4])
@pytest.mark.parametrize("case", WINO_CASES)
def test_conv3x3_winograd(ops, case, r):
    """Dense 3x3 conv as Winograd F(2x2, 3x3): input transform + ONE GEMM launch over the sixteen planes (per-plane
    weights) + output transform with BN / ReLU6 / residual, against F.conv2d on the CPU.  fp32 everywhere; the transforms
    add a few roundings per value (coefficients 0, +-1, +-1/2), hence 2e-4 instead of 8e-5."""
    n, h, w, cin, cout, act, use_res = case
    x = rnd((n, cin, h, w), 70, 2.0)
    wt = rnd((cout, cin, 3, 3), 71, 1.0 / np.sqrt(9 * cin))
    scale = rnd((cout,), 72) * 0.5 + 1.0
    bias = rnd((cout,), 73)
    res = rnd((n, cout, h, w), 74) if use_res else None
    ref = act_ref(F.conv2d(x, wt, padding=1) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1), act)
    if use_res:
        ref = ref + res
    got = ops.conv3x3_winograd(nhwc(x), wt, scale, bias, act=act, res=nhwc(res) if use_res else None, r=r)
    err = (nchw(got) - ref).abs().max().item()
    direct = ops.conv_gemm(nhwc(x), wt, scale, bias, act=act, res=nhwc(res) if use_res else None, prec="f32")
    print("winograd F(%dx%d) %s: max-abs vs F.conv2d %.2e (direct fp32 kernel: %.2e)" % (r, r, case, err, (nchw(direct) - ref).abs().max().item()))
    assert err <= (2e-4 if r == 2 else 6e-4), (case, r, err)       # F(4x4): coefficients up to 8, ~20x the rounding of direct fp32
    raw = ops.conv3x3_winograd(nhwc(x), wt, r=r)          # no BN (the hoisted W_x x_t of the ConvTWA layer)
    assert (nchw(raw) - F.conv2d(x, wt, padding=1)).abs().max().item() <= (2e-4 if r == 2 else 6e-4)


@pytest.mark.parametrize("r", [2, 4])
@pytest.mark.parametrize("case", [((45, 80), [(256, 12, 20), (128, 23, 40), (64, 45, 80)], 2), ((9, 13), [(8, 3, 4), (24, 5, 7)], 3),
                                  ((12, 20), [(32, 12, 20)], 1), ((23, 40), [(16, 1, 1), (12, 23, 40), (4, 12, 20)], 2)])
def test_conv3x3_winograd_virtual_concat_with_resize(ops, case, r):
    """The Winograd input transform reading a VIRTUAL concat: up to three tensors side by side along the channels, the ones on a
    smaller map resized on the fly (bilinear, align_corners=True) -- the SRF-Net head's `conv_last(cat[interpolate(x5),
    interpolate(x4), lv3])` (reference model.py:151-156) without resize launches or a concat buffer.  Against F.interpolate +
    torch.cat + F.conv2d on the CPU, and against the same conv on the materialised concat (the resize arithmetic is
    uavsal_bilinear_ac's: equal up to FMA contraction)."""
    from iip_uavsal_saliency_amd import _lib as L
    (h, w), segs, n = case
    xs = [rnd((n, c, sh, sw), 80 + i, 2.0) for i, (c, sh, sw) in enumerate(segs)]
    cin, cout = sum(c for c, _, _ in segs), 32
    wt = rnd((cout, cin, 3, 3), 90, 1.0 / np.sqrt(9 * cin))
    scale = rnd((cout,), 91) * 0.5 + 1.0
    bias = rnd((cout,), 92)
    cat = torch.cat([t if tuple(t.shape[2:]) == (h, w) else F.interpolate(t, size=(h, w), mode="bilinear", align_corners=True)
                     for t in xs], 1)
    ref = act_ref(F.conv2d(cat, wt, padding=1) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1), L.ACT_RELU6)
    got = ops.conv3x3_winograd([nhwc(t) for t in xs], wt, scale, bias, act=L.ACT_RELU6, r=r, size=(h, w))
    err = (nchw(got) - ref).abs().max().item()
    assert err <= (2e-4 if r == 2 else 6e-4), (case, r, err)
    whole = ops.conv3x3_winograd(nhwc(cat), wt, scale, bias, act=L.ACT_RELU6, r=r)
    assert (got - whole).abs().max().item() <= 5e-5, (case, r)


@pytest.mark.parametrize("r", [2, 4])
@pytest.mark.parametrize("shape", [(1, 45, 80), (2, 12, 20), (3, 9, 13)])
def test_twa_step_winograd(ops, shape, r):
    """ConvTWA step with the gate convolution through Winograd: the output transform applies sigmoid / convex update."""
    n, h, w = shape
    c = 256
    x = rnd((n, c, h, w), 75, 2.0)
    hp = rnd((n, c, h, w), 76, 2.0)
    wt = rnd((c, 2 * c, 3, 3), 77, 1.0 / np.sqrt(9 * 2 * c))
    gate = torch.sigmoid(F.conv2d(torch.cat([x, hp], 1), wt, padding=1))
    ref = gate * x + (1 - gate) * hp
    pre = ops.conv3x3_winograd(nhwc(x), wt[:, :c].contiguous(), r=r)
    got = ops.conv3x3_winograd(nhwc(hp), wt[:, c:].contiguous(), twa=(nhwc(x), pre), r=r)
    assert (nchw(got) - ref).abs().max().item() <= (2e-4 if r == 2 else 6e-4)


@pytest.mark.parametrize("prec", ["f32", "f16x3", "bf16x3"])
def test_twa_step(ops, prec):
    """ConvTWACell.forward (model_convlstm.py:276-292) with the x half of the conv hoisted."""
    n, c, h, w = 2, 256, 12, 20
    x = rnd((n, c, h, w), 41, 2.0)
    hp = rnd((n, c, h, w), 42, 2.0)
    wt = rnd((c, 2 * c, 3, 3), 43, 1.0 / np.sqrt(9 * 2 * c))
    gate = torch.sigmoid(F.conv2d(torch.cat([x, hp], 1), wt, padding=1))
    ref = gate * x + (1 - gate) * hp
    pre = ops.conv_gemm(nhwc(x), wt[:, :c].contiguous(), None, None, prec=prec)
    got = ops.twa_step(nhwc(x), nhwc(hp), pre, wt[:, c:].contiguous(), prec=prec)
    err = (nchw(got) - ref).abs().max().item()
    assert err <= TOL[prec] * 4.0, (prec, err)


def test_twa_step_stream_k(ops):
    """The ConvTWA step at the headline size (1 x 45 x 80: 228 tiles of 64 x 64) with the stream-K workspace:
    every tile is cut into 2-3 pieces, collected by the owner of its first K stage."""
    n, c, h, w = 1, 256, 45, 80
    x = rnd((n, c, h, w), 44, 2.0)
    hp = rnd((n, c, h, w), 45, 2.0)
    wt = rnd((c, 2 * c, 3, 3), 46, 1.0 / np.sqrt(9 * 2 * c))
    gate = torch.sigmoid(F.conv2d(torch.cat([x, hp], 1), wt, padding=1))
    ref = gate * x + (1 - gate) * hp
    pre = ops.conv_gemm(nhwc(x), wt[:, :c].contiguous(), None, None, prec="f32")
    whole = ops.twa_step(nhwc(x), nhwc(hp), pre, wt[:, c:].contiguous(), prec="f32")
    split = ops.twa_step(nhwc(x), nhwc(hp), pre, wt[:, c:].contiguous(), prec="f32", tile=4, stream_k=True)
    assert (split - whole).abs().max().item() <= TOL["f32"] * 4.0
    assert (nchw(split) - ref).abs().max().item() <= TOL["f32"] * 4.0


@pytest.mark.parametrize("shape", [(1, 45, 80), (2, 12, 20), (1, 23, 40), (3, 9, 13)])
def test_twa_step_f32_full_line_split_k(ops, shape):
    """fp32 ConvTWA step on the kernel with 32-float K stages (tile 8) with K split over up to 8 workgroups per
    128 x 128 tile (one clip at 45 x 80: 58 tiles x 72 stages -> 464 shares); the shares meet in the reduce launch, which
    applies the ConvTWA update.  Against F.conv2d on the CPU, the whole-tile launch (tile 4) and its own second run."""
    n, h, w = shape
    c = 256
    x = rnd((n, c, h, w), 50, 2.0)
    hp = rnd((n, c, h, w), 51, 2.0)
    wt = rnd((c, 2 * c, 3, 3), 52, 1.0 / np.sqrt(9 * 2 * c))
    gate = torch.sigmoid(F.conv2d(torch.cat([x, hp], 1), wt, padding=1))
    ref = gate * x + (1 - gate) * hp
    pre = ops.conv_gemm(nhwc(x), wt[:, :c].contiguous(), None, None, prec="f32")
    whole = ops.twa_step(nhwc(x), nhwc(hp), pre, wt[:, c:].contiguous(), prec="f32", tile=4)
    for tile in (8, 10, 11, 0):     # 8: shares + reduce launch; 10: flat pipeline, the last share to arrive reduces;
                                    # 11: 64 x 64 tiles, shares over workgroups, reduced in the launch; 0: the default
        split = ops.twa_step(nhwc(x), nhwc(hp), pre, wt[:, c:].contiguous(), prec="f32", tile=tile, stream_k=True)
        again = ops.twa_step(nhwc(x), nhwc(hp), pre, wt[:, c:].contiguous(), prec="f32", tile=tile, stream_k=True)
        assert torch.equal(split, again), tile                       # fixed summation order, whoever reduces
        assert (split - whole).abs().max().item() <= TOL["f32"] * 4.0, tile
        assert (nchw(split) - ref).abs().max().item() <= TOL["f32"] * 4.0, tile


@pytest.mark.parametrize("case", [(2, 12, 20, 1920, 256, 1, 1, True), (1, 23, 40, 960, 160, 1, 0, False),
                                  (8, 12, 20, 1024, 256, 1, 1, False), (2, 12, 20, 256, 256, 9, 1, True)])
def test_conv_f32_full_line_split_k(ops, case):
    """Affine convs with few 128 x 128 tiles and a long K on tile 8 with the workspace: K split + reduce launch."""
    n, h, w, cin, cout, taps, act, use_res = case
    kk = 3 if taps == 9 else 1
    x = rnd((n, cin, h, w), 53, 2.0)
    wt = rnd((cout, cin, kk, kk), 54, 1.0 / np.sqrt(cin * taps))
    scale = rnd((cout,), 55) * 0.5 + 1.0
    bias = rnd((cout,), 56)
    res = rnd((n, cout, h, w), 57) if use_res else None
    ref = act_ref(F.conv2d(x, wt, padding=kk // 2) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1), act)
    if use_res:
        ref = ref + res
    for tile in (8, 10, 11):
        got = ops.conv_gemm(nhwc(x), wt, scale, bias, act=act, res=nhwc(res) if use_res else None, prec="f32", tile=tile,
                            stream_k=True)
        assert (nchw(got) - ref).abs().max().item() <= TOL["f32"] * 4.0, (tile, case)


@pytest.mark.parametrize("tile", [10, 11])
def test_k_split_tickets_survive_stale_streamk_flags(ops, tile):
    """the ticket counters of the in-launch K-share reduction have their own region of the workspace's flag block: flags a
    stream-K wait that gave up left behind (first region, re-zeroed by the host only before the NEXT forward) must not move a later
    launch's tickets.  Same result, bit for bit, with a clean and with a dirtied flag region."""
    n, h, w, cin, cout = 2, 12, 20, 1920, 256
    x = rnd((n, cin, h, w), 153, 2.0)
    wt = rnd((cout, cin, 1, 1), 154, 1.0 / np.sqrt(cin))
    scale = rnd((cout,), 155) * 0.5 + 1.0
    bias = rnd((cout,), 156)
    ref = act_ref(F.conv2d(x, wt) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1), 1)
    clean = ops.conv_gemm(nhwc(x), wt, scale, bias, act=1, prec="f32", tile=tile, stream_k=True)
    dirty = ops.conv_gemm(nhwc(x), wt, scale, bias, act=1, prec="f32", tile=tile, stream_k=True, _stale_streamk_flags=True)
    assert torch.equal(clean, dirty), tile
    assert (nchw(dirty) - ref).abs().max().item() <= TOL["f32"] * 4.0, tile


@pytest.mark.parametrize("tile", [8, 9, 10, 11])
@pytest.mark.parametrize("case", [(2, 12, 20, 256, 1536, 1, 1, False), (1, 23, 40, 1536, 256, 1, 0, True),
                                  (3, 7, 5, 320, 256, 1, 1, False), (1, 45, 80, 32, 256, 1, 1, True),
                                  (2, 13, 17, 96, 200, 1, 0, False), (1, 12, 20, 64, 96, 9, 1, False),
                                  (2, 9, 13, 448, 256, 9, 1, False), (1, 45, 80, 256, 256, 9, 0, True)])
def test_conv_f32_full_line_tiles(ops, tile, case):
    """The fp32 kernels with 32-float (one cache line per row) K stages: 128 x 128 (8), 256 x 128 (9), the flat
    pipeline (10) and 64 x 64 (11; with the workspace: K shares over workgroups), 1x1 and 3x3 (weights packed with 32-channel K blocks), ragged M / N, residual."""
    n, h, w, cin, cout, taps, act, use_res = case
    kk = 3 if taps == 9 else 1
    x = rnd((n, cin, h, w), 58, 2.0)
    wt = rnd((cout, cin, kk, kk), 59, 1.0 / np.sqrt(cin * taps))
    scale = rnd((cout,), 60) * 0.5 + 1.0
    bias = rnd((cout,), 61)
    res = rnd((n, cout, h, w), 62) if use_res else None
    ref = act_ref(F.conv2d(x, wt, padding=kk // 2) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1), act)
    if use_res:
        ref = ref + res
    got = ops.conv_gemm(nhwc(x), wt, scale, bias, act=act, res=nhwc(res) if use_res else None, prec="f32", tile=tile)
    assert (nchw(got) - ref).abs().max().item() <= TOL["f32"] * 4.0, (tile, case)


@pytest.mark.parametrize("prec", ["f16x3", "bf16x3"])
@pytest.mark.parametrize("shape", [(1, 45, 80), (2, 12, 20), (1, 23, 40)])
def test_twa_step_split_k(ops, prec, shape):
    """Split-16-bit ConvTWA step with the workspace: few 64 x 64 tiles and 72 K steps -> K is split over 2-4
    workgroups per tile, the shares meet in the reduce launch (which also does the ConvTWA update)."""
    n, h, w = shape
    c = 256
    x = rnd((n, c, h, w), 47, 2.0)
    hp = rnd((n, c, h, w), 48, 2.0)
    wt = rnd((c, 2 * c, 3, 3), 49, 1.0 / np.sqrt(9 * 2 * c))
    gate = torch.sigmoid(F.conv2d(torch.cat([x, hp], 1), wt, padding=1))
    ref = gate * x + (1 - gate) * hp
    pre = ops.conv_gemm(nhwc(x), wt[:, :c].contiguous(), None, None, prec=prec)
    whole = ops.twa_step(nhwc(x), nhwc(hp), pre, wt[:, c:].contiguous(), prec=prec)
    split = ops.twa_step(nhwc(x), nhwc(hp), pre, wt[:, c:].contiguous(), prec=prec, stream_k=True)
    assert (split - whole).abs().max().item() <= TOL[prec] * 4.0
    assert (nchw(split) - ref).abs().max().item() <= TOL[prec] * 4.0


@pytest.mark.parametrize("prec", ["f16x3", "bf16x3"])
def test_conv1x1_small_map_split_k(ops, prec):
    """1x1 projection with a long K on a small map (ASPP: 1920 -> 256 at 12x20) with the workspace: K split + reduce launch."""
    x = rnd((2, 1920, 12, 20), 56, 2.0)
    wt = rnd((256, 1920, 1, 1), 57, 1.0 / np.sqrt(1920))
    sc, bi = rnd((256,), 58) * 0.5 + 1.0, rnd((256,), 59)
    res = rnd((2, 256, 12, 20), 60)
    ref = F.conv2d(x, wt) * sc.view(1, -1, 1, 1) + bi.view(1, -1, 1, 1) + res
    got = ops.conv_gemm(nhwc(x), wt, sc, bi, res=nhwc(res), prec=prec, tile=4, stream_k=True)
    whole = ops.conv_gemm(nhwc(x), wt, sc, bi, res=nhwc(res), prec=prec, tile=4)
    assert (got - whole).abs().max().item() <= TOL[prec] * 4.0
    assert (nchw(got) - ref).abs().max().item() <= TOL[prec] * 4.0


@pytest.mark.parametrize("case", [(2, 13, 17, 1024, 96), (1, 7, 9, 768, 36), (3, 5, 5, 2048, 200)])
def test_conv1x1_split_k_ragged(ops, case):
    """K split with row counts that are no tile multiple and output widths that leave a partial last 64-wide tile."""
    n, h, w, cin, cout = case
    x = rnd((n, cin, h, w), 61, 2.0)
    wt = rnd((cout, cin, 1, 1), 62, 1.0 / np.sqrt(cin))
    sc, bi = rnd((cout,), 63) * 0.5 + 1.0, rnd((cout,), 64)
    ref = torch.clamp(F.conv2d(x, wt) * sc.view(1, -1, 1, 1) + bi.view(1, -1, 1, 1), 0, 6)
    from iip_uavsal_saliency_amd import _lib as L
    got = ops.conv_gemm(nhwc(x), wt, sc, bi, act=L.ACT_RELU6, prec="f16x3", tile=4, stream_k=True)
    assert (nchw(got) - ref).abs().max().item() <= TOL["f16x3"] * 4.0


def test_conv3x3_small_map_split_k(ops):
    """A plain 3x3 conv (BN + ReLU6 + residual) on a small map in f16x3 with the workspace: same split path, affine epilogue."""
    x = rnd((2, 64, 23, 40), 51, 2.0)
    wt = rnd((64, 64, 3, 3), 52, 1.0 / np.sqrt(9 * 64))
    sc, bi = rnd((64,), 53) * 0.5 + 1.0, rnd((64,), 54)
    res = rnd((2, 64, 23, 40), 55)
    ref = torch.clamp(F.conv2d(x, wt, padding=1) * sc.view(1, -1, 1, 1) + bi.view(1, -1, 1, 1), 0, 6) + res
    from iip_uavsal_saliency_amd import _lib as L
    got = ops.conv_gemm(nhwc(x), wt, sc, bi, act=L.ACT_RELU6, res=nhwc(res), prec="f16x3", tile=4, stream_k=True)
    assert (nchw(got) - ref).abs().max().item() <= TOL["f16x3"] * 4.0


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
def test_convlstm_step_vs_reference_golden(ops, prec, golden_dir):
    """ConvLSTMCell.forward: the golden was produced by the reference's own model_convlstm.py."""
    import os
    from iip_uavsal_saliency_amd import synth
    g = np.load(os.path.join(golden_dir, "convlstm_step.npz"))
    hid, (h, w), seed = int(g["hid"]), g["hw"], int(g["seed"])
    wgt = torch.from_numpy(synth.synth_tensor("convlstm.rnn_conv.weight", (4 * hid, 2 * hid, 3, 3), seed))
    mk = lambda nm: torch.from_numpy(synth.hash_normal(nm, hid * h * w, seed).astype(np.float32)).view(1, hid, h, w)
    x, hp, cp = mk("convlstm.x"), mk("convlstm.h"), mk("convlstm.c")
    hn, cn = ops.lstm_step(nhwc(x), nhwc(hp), nhwc(cp), wgt, prec=prec)
    assert np.abs(nchw(hn).numpy() - g["h_next"]).max() <= TOL[prec] * 4
    assert np.abs(nchw(cn).numpy() - g["c_next"]).max() <= TOL[prec] * 4


def test_convlstm_step_256(ops):
    from oracle.uavsal_ref import convlstm_cell_step
    n, c, h, w = 2, 256, 12, 20
    x, hp, cp = rnd((n, c, h, w), 141), rnd((n, c, h, w), 142), rnd((n, c, h, w), 143)
    wt = rnd((4 * c, 2 * c, 3, 3), 144, 1.0 / np.sqrt(9 * 2 * c))
    rh, rc = convlstm_cell_step(wt, x, hp, cp)
    hn, cn = ops.lstm_step(nhwc(x), nhwc(hp), nhwc(cp), wt)
    assert (nchw(hn) - rh).abs().max().item() <= 1e-4 and (nchw(cn) - rc).abs().max().item() <= 1e-4


FUSED_CASES = [(2, 12, 20, 384, 64, 1, True), (1, 45, 80, 1536, 256, 1, False), (2, 23, 41, 96, 24, 2, False),
               (1, 9, 13, 48, 64, 1, False), (2, 45, 80, 1152, 64, 1, False), (1, 45, 80, 192, 64, 2, False)]


@pytest.mark.parametrize("prec", ["f32", "f16x3", "bf16x3"])
@pytest.mark.parametrize("case", FUSED_CASES)
def test_fused_depthwise_projection(ops, prec, case):
    """dw3x3+BN+ReLU6 -> 1x1+BN (+res) in one launch == the two reference convs of dwBlock (model.py:92-95)."""
    n, h, w, c, cout, stride, use_res = case
    e = rnd((n, c, h, w), 151, 3.0).clamp(0, 6)
    wd = rnd((c, 1, 3, 3), 152, 0.4)
    sd, bd = rnd((c,), 153) * 0.5 + 1.0, rnd((c,), 154)
    wp = rnd((cout, c, 1, 1), 155, 1.0 / np.sqrt(c))
    sp, bp = rnd((cout,), 156) * 0.5 + 1.0, rnd((cout,), 157)
    dmid = torch.clamp(F.conv2d(e, wd, stride=stride, padding=1, groups=c) * sd.view(1, -1, 1, 1) + bd.view(1, -1, 1, 1), 0, 6)
    ref = F.conv2d(dmid, wp) * sp.view(1, -1, 1, 1) + bp.view(1, -1, 1, 1)
    res = rnd(tuple(ref.shape), 158) if use_res else None
    if use_res:
        ref = ref + res
    got = ops.conv_gemm(nhwc(e), wp, sp, bp, res=nhwc(res) if use_res else None, prec=prec, dw=(wd, sd, bd, stride))
    assert tuple(got.shape) == (n, ref.shape[2], ref.shape[3], cout)
    err = (nchw(got) - ref).abs().max().item()
    assert err <= TOL[prec] * 8.0, (case, prec, err)


# LDS-halo depthwise -> projection kernel (dwproj_kernel, fp32 and split-fp16): every instance (output tile 256 / 128 / 64 / 32), maps
# that are not multiples of the 8 x 16 patch, one row / one column maps, several images (tile walk), residual,
# sigmoid + Cout = 1 (scalar store path: the decoder's last launch), output / residual as channel slices
DWPROJ_CASES = [
    # n, h, w, hidden, cout, act, residual, sliced output
    (1, 45, 80, 1536, 256, 0, True, False), (2, 45, 80, 320, 1, 2, False, False), (3, 13, 21, 96, 192, 1, False, False),
    (2, 9, 17, 48, 128, 0, True, True), (1, 8, 16, 64, 64, 0, False, False), (2, 1, 37, 32, 40, 1, False, True),
    (1, 23, 1, 16, 24, 0, False, False), (17, 7, 5, 80, 32, 2, False, False), (1, 90, 160, 64, 256, 0, False, False),
    (40, 17, 33, 16, 300, 0, True, False),
    # narrow outputs with few tiles and a long K walk: K is split over 2-4 workgroups per tile + the reduce launch
    (1, 16, 32, 768, 1, 2, False, False), (2, 9, 20, 384, 40, 1, True, True), (1, 8, 16, 1024, 64, 0, True, False),
    (3, 20, 20, 400, 33, 0, False, False),
]


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
@pytest.mark.parametrize("case", DWPROJ_CASES)
def test_depthwise_projection_lds_halo(ops, case, prec):
    """dwBlock tail (model.py:92-95) in one launch with the halo tile in LDS == the two convs in fp32 torch."""
    from iip_uavsal_saliency_amd import _lib as L
    import ctypes as C
    n, h, w, c, cout, act, use_res, sliced = case
    e = rnd((n, c, h, w), 251, 3.0).clamp(0, 6)
    wd = rnd((c, 1, 3, 3), 252, 0.4)
    sd, bd = rnd((c,), 253) * 0.5 + 1.0, rnd((c,), 254)
    wp = rnd((cout, c, 1, 1), 255, 1.0 / np.sqrt(c))
    sp, bp = rnd((cout,), 256) * 0.5 + 1.0, rnd((cout,), 257)
    dmid = torch.clamp(F.conv2d(e, wd, padding=1, groups=c) * sd.view(1, -1, 1, 1) + bd.view(1, -1, 1, 1), 0, 6)
    ref = F.conv2d(dmid, wp) * sp.view(1, -1, 1, 1) + bp.view(1, -1, 1, 1)
    ref = torch.clamp(ref, 0, 6) if act == 1 else (torch.sigmoid(ref) if act == 2 else ref)
    res = rnd(tuple(ref.shape), 258) if use_res else None
    if use_res:
        ref = ref + res
    d = L.ConvDesc()
    d.n_img, d.H, d.W, d.Cin, d.Cout, d.taps, d.prec, d.epi = n, h, w, c, cout, 1, L.PREC[prec], L.EPI_AFFINE
    d.dw_w9c, d.dw_stride, d.out = 1 << 20, 1, 1 << 20
    inst = int(L.load().uavsal_conv_dwproj(C.byref(d)))
    assert inst == (256 if cout > 128 else 128 if cout > 64 else 64 if cout > 32 else 32)
    pad = 8 if sliced else 0            # the output (and residual) are channel slices of wider NHWC buffers
    dev = nhwc(e).device
    outbuf = torch.full((n, h, w, cout + 2 * pad), 7.0, device=dev)
    out = outbuf[..., pad:pad + cout]
    rbuf = None
    if use_res:
        rbuf = torch.zeros((n, h, w, cout + 2 * pad), device=dev)
        rbuf[..., pad:pad + cout] = nhwc(res)
    got = ops.conv_gemm(nhwc(e), wp, sp, bp, act=act, res=rbuf[..., pad:pad + cout] if use_res else None, out=out,
                        prec=prec, dw=(wd, sd, bd, 1))
    err = (nchw(got.contiguous()) - ref).abs().max().item()
    assert err <= (2e-5 if prec == "f32" else TOL[prec]) * max(1.0, ref.abs().max().item()), (case, prec, err)
    if sliced:                          # nothing outside the slice was written
        assert (outbuf[..., :pad] == 7.0).all() and (outbuf[..., pad + cout:] == 7.0).all()


def _dwproj_fuzz_cases():
    """Seeded random shapes: K walks of 1..7 steps (lead-in / tail conditions of the pipelined loop), output widths that
    leave a partial last tile or a weight panel narrower than the tile, odd maps, several images."""
    g = np.random.RandomState(7)
    cases = []
    for i in range(16):
        hid = 16 * int(g.choice([1, 2, 3, 4, 5, 6, 7, 12]))
        cout = int(g.choice([1, 3, 17, 32, 33, 64, 65, 100, 129, 200, 257, 320]))
        cases.append((int(g.randint(1, 4)), int(g.randint(1, 30)), int(g.randint(1, 40)), hid, cout, int(g.randint(0, 3)),
                      bool(g.randint(0, 2))))
    return cases + [(1, 9, 9, 48, 200, 0, False), (1, 9, 9, 48, 257, 0, True)]


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
@pytest.mark.parametrize("case", _dwproj_fuzz_cases())
def test_depthwise_projection_lds_halo_random_shapes(ops, case, prec):
    n, h, w, c, cout, act, use_res = case
    e = rnd((n, c, h, w), 271, 3.0).clamp(0, 6)
    wd, sd, bd = rnd((c, 1, 3, 3), 272, 0.4), rnd((c,), 273) * 0.5 + 1.0, rnd((c,), 274)
    wp, sp, bp = rnd((cout, c, 1, 1), 275, 1.0 / np.sqrt(c)), rnd((cout,), 276) * 0.5 + 1.0, rnd((cout,), 277)
    dmid = torch.clamp(F.conv2d(e, wd, padding=1, groups=c) * sd.view(1, -1, 1, 1) + bd.view(1, -1, 1, 1), 0, 6)
    ref = F.conv2d(dmid, wp) * sp.view(1, -1, 1, 1) + bp.view(1, -1, 1, 1)
    ref = torch.clamp(ref, 0, 6) if act == 1 else (torch.sigmoid(ref) if act == 2 else ref)
    res = rnd(tuple(ref.shape), 278) if use_res else None
    if use_res:
        ref = ref + res
    got = ops.conv_gemm(nhwc(e), wp, sp, bp, act=act, res=nhwc(res) if use_res else None, prec=prec, dw=(wd, sd, bd, 1))
    err = (nchw(got) - ref).abs().max().item()
    assert err <= (2e-5 if prec == "f32" else TOL[prec]) * max(1.0, ref.abs().max().item()), (case, prec, err)


def test_depthwise_projection_split_shadow(ops):
    """f16x3: the fused launch also writes the split shadow of its output (what the next GEMM stages by LDS-DMA)."""
    n, h, w, c, cout = 2, 19, 35, 96, 128
    e = rnd((n, c, h, w), 261, 3.0).clamp(0, 6)
    wd, sd, bd = rnd((c, 1, 3, 3), 262, 0.4), rnd((c,), 263) * 0.5 + 1.0, rnd((c,), 264)
    wp, sp, bp = rnd((cout, c, 1, 1), 265, 1.0 / np.sqrt(c)), rnd((cout,), 266) * 0.5 + 1.0, rnd((cout,), 267)
    out, shadow = ops.conv_gemm(nhwc(e), wp, sp, bp, prec="f16x3", dw=(wd, sd, bd, 1), split_out=True)
    assert (ops.merge_shadow(shadow) - out).abs().max().item() <= 3.0 * 2 ** -20 * max(1.0, out.abs().max().item())


DW_CASES = [
    # n, h, w, c, stride, dilation
    (2, 9, 13, 48, 1, 1), (1, 12, 20, 120, 1, 1), (2, 45, 80, 1536, 1, 1), (1, 23, 41, 96, 2, 1),
    (2, 45, 80, 1536, 2, 1), (1, 180, 320, 32, 1, 1), (2, 12, 20, 1920, 1, 6), (2, 12, 20, 1920, 1, 12),
    (2, 12, 20, 1920, 1, 18), (1, 5, 3, 144, 2, 1), (1, 1, 1, 64, 1, 1), (1, 2, 2, 64, 2, 1),
    # whole-map LDS kernel: last channel slab narrower than the slab, 8 images (slab 32), odd map, and a map
    # too large for LDS (generic dilated kernel)
    (1, 12, 20, 120, 1, 2), (8, 12, 20, 960, 1, 6), (3, 23, 40, 96, 1, 3), (1, 45, 80, 64, 1, 2), (2, 7, 5, 20, 1, 4),
    # thousands of (image, slab) workgroups of the whole-map kernel (XCD-renumbered blocks), last slab narrower than 32 channels
    (40, 12, 20, 3824, 1, 6), (36, 23, 40, 1840, 1, 6),
]


@pytest.mark.parametrize("case", DW_CASES)
def test_depthwise(ops, case):
    n, h, w, c, stride, dil = case
    x = rnd((n, c, h, w), 51, 2.0)
    wt = rnd((c, 1, 3, 3), 52, 0.4)
    scale = rnd((c,), 53) * 0.5 + 1.0
    bias = rnd((c,), 54)
    ref = torch.clamp(F.conv2d(x, wt, stride=stride, padding=dil, dilation=dil, groups=c)
                      * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1), 0, 6)
    got = ops.dw3x3(nhwc(x), wt, scale, bias, stride=stride, dilation=dil)
    assert tuple(got.shape) == (n, ref.shape[2], ref.shape[3], c)
    err = (nchw(got) - ref).abs().max().item()
    assert err <= 1e-5, (case, err)       # 9 fp32 fmas per output, order may differ


@pytest.mark.parametrize("case", [(8, 12, 20, 3 * 1920, (6, 12, 18)), (2, 9, 13, 192, (2, 3, 5)), (1, 45, 80, 128, (1, 4)), (2, 130, 140, 96, (3, 7, 1)),
                                  (48, 9, 13, 3 * 960, (6, 2, 3)),         # (4320 workgroups of the whole-map kernel)
                                  # maps too big for 32-channel whole-map slabs: dw3x3_rowclass_kernel (rows of one residue class per
                                  # workgroup).  The 1/32 level of 720x1280 inputs; narrow groups (32-channel slabs); a dilation beyond
                                  # the map (every row its own class) next to a small one; one group only
                                  (4, 23, 40, 3 * 1920, (6, 12, 18)), (2, 23, 40, 3 * 96, (6, 12, 18)), (3, 25, 33, 2 * 64, (30, 5)),
                                  (2, 20, 30, 64, (40,)),
                                  # a group whose classes do not fit (dilation 2: 12 rows): the whole launch stays on the other kernels
                                  (1, 23, 40, 4 * 32, (2, 6, 12, 18))])
def test_depthwise_channel_groups_with_their_own_dilation(ops, case):
    """uavsal_dw_desc.dil_group_c: the dilated branches of one map in one launch (the three ASPP depthwise convs on the slices of
    their merged expand, model.py:125-127, 142-147) == one torch depthwise conv per group.  Whole-map LDS kernel where the map
    fits, the per-pixel kernel otherwise (130x140)."""
    n, h, w, c, dils = case
    x = rnd((n, c, h, w), 171, 2.0)
    wd = rnd((c, 1, 3, 3), 172, 0.4)
    sd, bd = rnd((c,), 173) * 0.5 + 1.0, rnd((c,), 174)
    g = c // len(dils)
    ref = torch.cat([F.conv2d(x[:, i * g:(i + 1) * g], wd[i * g:(i + 1) * g], padding=d, dilation=d, groups=g) for i, d in enumerate(dils)], 1)
    ref = torch.clamp(ref * sd.view(1, -1, 1, 1) + bd.view(1, -1, 1, 1), 0, 6)
    got = ops.dw3x3(nhwc(x), wd, sd, bd, dil_groups=list(dils))
    assert (nchw(got) - ref).abs().max().item() <= 2e-5, case
    from iip_uavsal_saliency_amd import _lib as L
    d = L.DwDesc()
    d.n_img, d.H, d.W, d.C, d.stride, d.dilation, d.dil_group_c = n, h, w, c, 1, 1, g
    for i, dd in enumerate(dils):
        d.dil_groups[i] = dd
    variant = L.load().uavsal_dw_variant(ctypes.byref(d))
    assert (variant == 2048) == (case in ((4, 23, 40, 3 * 1920, (6, 12, 18)), (2, 23, 40, 3 * 96, (6, 12, 18)), (3, 25, 33, 2 * 64, (30, 5)),
                                          (2, 20, 30, 64, (40,)))), (case, variant)
    with pytest.raises(RuntimeError):
        ops.dw3x3(nhwc(x), wd, sd, bd, stride=2, dil_groups=list(dils))


@pytest.mark.parametrize("act", [0, 2])
@pytest.mark.parametrize("case", [(2, 45, 80, 1536), (1, 7, 9, 256), (3, 13, 17, 512), (1, 4, 4, 2048), (1, 1, 1, 256)])
def test_depthwise_dot_projection(ops, case, act):
    """uavsal_dw3x3_dot: depthwise 3x3 + BN + ReLU6 -> 1x1 projection to ONE channel + BN + activation (the tail of conv_out_st,
    reference model.py:92-96, 333-334, 372-373) against the three torch ops; deterministic (fixed summation order)."""
    n, h, w, c = case
    x = rnd((n, c, h, w), 161, 2.0)
    wd = rnd((c, 1, 3, 3), 162, 0.4)
    sd, bd = rnd((c,), 163) * 0.5 + 1.0, rnd((c,), 164)
    w2 = rnd((1, c, 1, 1), 165, 2.0 / np.sqrt(c))
    s2, b2 = 1.3, -0.2
    d = torch.clamp(F.conv2d(x, wd, padding=1, groups=c) * sd.view(1, -1, 1, 1) + bd.view(1, -1, 1, 1), 0, 6)
    ref = F.conv2d(d, w2) * s2 + b2
    if act == 2:
        ref = torch.sigmoid(ref)
    got = ops.dw3x3_dot(nhwc(x), wd, sd, bd, w2, s2, b2, act=act)
    again = ops.dw3x3_dot(nhwc(x), wd, sd, bd, w2, s2, b2, act=act)
    assert torch.equal(got, again)
    assert (nchw(got) - ref).abs().max().item() <= 3e-5, case
    with pytest.raises(RuntimeError):
        ops.dw3x3_dot(nhwc(rnd((1, 384, 5, 5), 166)), rnd((384, 1, 3, 3), 167), rnd((384,), 168), rnd((384,), 169),
                      rnd((1, 384, 1, 1), 170), 1.0, 0.0)          # one workgroup = all channels of a patch: C % 256


@pytest.mark.parametrize("size", [(2, 36, 64), (1, 45, 81), (2, 7, 9), (1, 21, 300), (2, 360, 640), (1, 1, 1), (1, 17, 129)])
def test_stem(ops, size):
    """(the kernel works on 4 x 64 output tiles: several tiles per row, ragged last tiles, maps smaller than a tile)"""
    n, H, W = size
    x = rnd((n, 3, H, W), 61, 2.0)
    wt = rnd((32, 3, 3, 3), 62, 0.3)
    scale = rnd((32,), 63) * 0.5 + 1.0
    bias = rnd((32,), 64)
    ref = torch.clamp(F.conv2d(x, wt, stride=2, padding=1) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1), 0, 6)
    got = ops.stem_conv(x.cuda(), wt, scale, bias)
    assert (nchw(got) - ref).abs().max().item() <= 2e-5


def test_stem_uint8_normalisation(ops):
    """uint8 frames normalised on load == normalize_data (utils_data.py:43-65) then the fp32 stem."""
    from iip_uavsal_saliency_amd import synth
    u8 = synth.synth_frames_u8(2, 24, 300)
    wt = rnd((32, 3, 3, 3), 62, 0.3)
    scale = rnd((32,), 63) * 0.5 + 1.0
    bias = rnd((32,), 64)
    a = ops.stem_conv(torch.from_numpy(synth.normalize_frames(u8)).cuda(), wt, scale, bias)
    b = ops.stem_conv(torch.from_numpy(u8).cuda(), wt, scale, bias)
    assert (a - b).abs().max().item() <= 2e-5


@pytest.mark.parametrize("case", [(2, 12, 20, 256, 45, 80), (1, 23, 40, 128, 45, 80), (2, 3, 5, 64, 12, 20), (1, 1, 1, 8, 4, 4)])
def test_bilinear_align_corners(ops, case):
    n, hi, wi, c, ho, wo = case
    x = rnd((n, c, hi, wi), 71)
    ref = F.interpolate(x, size=(ho, wo), mode="bilinear", align_corners=True)
    got = ops.bilinear_ac(nhwc(x), ho, wo)
    assert (nchw(got) - ref).abs().max().item() <= 2e-6


def test_bilinear_context_maps(ops):
    """`cb_cxt.repeat(T,1,1,1)` tiling (model.py:361) and the per-clip map of forward_clips."""
    B, T, c = 3, 4, 64
    x = rnd((B, c, 3, 5), 72)
    up = F.interpolate(x, size=(12, 20), mode="bilinear", align_corners=True)
    tiled = up.repeat(T, 1, 1, 1)                                 # frame k <- chunk k % B
    got = ops.bilinear_ac(nhwc(x), 12, 20, n_out=B * T, src_mod=B, src_div=1)
    assert (nchw(got) - tiled).abs().max().item() <= 2e-6
    per_clip = up.repeat_interleave(T, 0)                         # frame (c,t) <- clip c
    got = ops.bilinear_ac(nhwc(x), 12, 20, n_out=B * T, src_mod=B * T, src_div=T)
    assert (nchw(got) - per_clip).abs().max().item() <= 2e-6


@pytest.mark.parametrize("n,L", [(5, 5), (6, 3), (2, 2), (8, 4)])
def test_temporal_differences(ops, n, L):
    from oracle.uavsal_ref import temporal_differences
    x = rnd((n, 32, 6, 7), 81)
    ref = torch.cat([temporal_differences(x[i:i + L]) for i in range(0, n, L)], 0)
    got = ops.tdiff(nhwc(x), L)
    assert torch.equal(nchw(got), ref)     # single fp32 subtraction: bit-exact


def test_time_sum(ops):
    x = rnd((8, 256, 5, 6), 91)
    ref = x.view(2, 4, 256, 5, 6).sum(1)
    got = ops.tsum(nhwc(x), 4)
    assert (nchw(got) - ref).abs().max().item() <= 1e-6


@pytest.mark.parametrize("c,hw", [(8, (45, 80)), (20, (12, 20)), (256, (9, 13)), (3, (5, 5))])
def test_layout_roundtrip(ops, c, hw):
    x = rnd((2, c, hw[0], hw[1]), 95)
    y = ops.to_nhwc(x.cuda())
    assert torch.equal(y.cpu(), x.permute(0, 2, 3, 1).contiguous())
    z = ops.to_nchw(y)
    assert torch.equal(z.cpu(), x)


def test_f16x3_saturates_instead_of_overflowing(ops):
    """|x| * 16 beyond the fp16 range clips (documented in uavsal_hip.h) -- never inf/nan."""
    x = torch.full((1, 32, 4, 4), 1.0e5)
    wt = torch.full((32, 32, 1, 1), 0.01)
    got = ops.conv_gemm(nhwc(x), wt, None, None, prec="f16x3")
    assert torch.isfinite(got).all()


def test_rejects_bad_arguments(ops):
    x = torch.zeros((1, 4, 4, 6), device="cuda")          # Cin % 4 != 0
    with pytest.raises(RuntimeError):
        ops.conv_gemm(x, torch.zeros(8, 6, 1, 1), None, None)
    with pytest.raises(RuntimeError):                      # 3x3 needs Cin % 32 == 0
        ops.conv_gemm(torch.zeros((1, 4, 4, 16), device="cuda"), torch.zeros(8, 16, 3, 3), None, None)
    with pytest.raises(RuntimeError):                      # one frame: reference raises too (model.py:194)
        ops.tdiff(torch.zeros((1, 4, 4, 32), device="cuda"), 1)


# ---- pre-split (LDS-DMA) fp16 GEMM path: uavsal_conv_desc.a_split, out_split -----------------------------
SPLIT_CASES = [
    # n, h, w, cin, cout, taps, act, res, tile
    (2, 12, 20, 256, 1536, 1, 1, False, 1),     # expand, 128x128 (2 x 32 KB ring)
    (2, 12, 20, 256, 1536, 1, 1, False, 5),     # 128x256, 8 waves, 3-stage ring
    (2, 23, 40, 256, 1536, 1, 1, False, 6),     # 256x256, 8 waves, M tail (1840 rows)
    (1, 23, 40, 1536, 256, 1, 0, True, 5),      # projection with residual, long K
    (1, 23, 40, 1536, 256, 1, 0, True, 6),
    (3, 7, 5, 320, 256, 1, 1, False, 1),        # M = 105 < one tile, Cin = 320 (10 K steps)
    (2, 9, 13, 96, 64, 1, 0, False, 1),         # Cout = 64 < tile width, 3 K steps
    (1, 12, 20, 448, 256, 9, 1, False, 5),      # conv_last shape: 3x3, zero padding through the zero page
    (2, 9, 11, 64, 96, 9, 1, True, 1),          # 3x3, ragged everything
    (1, 45, 80, 256, 256, 9, 0, False, 6),      # 3x3 on the 256x256 tile
    (2, 45, 80, 32, 256, 1, 1, False, 1),       # one K step
]


@pytest.mark.parametrize("case", SPLIT_CASES)
def test_conv_presplit_lds_dma(ops, case):
    """The pre-split path == F.conv2d fp32 on CPU at the f16x3 tolerance, and the split shadow it writes for
    its own output reconstructs that output to fp16x2 precision."""
    n, h, w, cin, cout, taps, act, use_res, tile = case
    k = 3 if taps == 9 else 1
    x = rnd((n, cin, h, w), 81, 2.0)
    wt = rnd((cout, cin, k, k), 82, 1.0 / np.sqrt(cin * taps))
    scale = rnd((cout,), 83) * 0.5 + 1.0
    bias = rnd((cout,), 84)
    res = rnd((n, cout, h, w), 85) if use_res else None
    ref = act_ref(F.conv2d(x, wt, padding=k // 2) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1), act)
    if use_res:
        ref = ref + res
    got, sp = ops.conv_gemm(nhwc(x), wt, scale, bias, act=act, res=nhwc(res) if use_res else None,
                            prec="f16x3", tile=tile, split_in=True, split_out=True)
    mag = max(1.0, ref.abs().max().item())
    err = (nchw(got) - ref).abs().max().item()
    assert err <= TOL["f16x3"] * 8.0 * mag, (case, err)
    # shadow of the output: hi + lo == 16 * out up to the two roundings (2^-21 relative) 
    rec = ops.merge_shadow(sp)
    serr = (rec - got).abs().max().item()
    assert serr <= 2e-6 * mag, (case, serr)
    # and it matches the register-staged kernel on the same operands to accumulation-order noise
    old = ops.conv_gemm(nhwc(x), wt, scale, bias, act=act, res=nhwc(res) if use_res else None, prec="f16x3",
                        tile=tile if tile in (1, 5, 6) else 0)
    assert (old - got).abs().max().item() <= 2e-5 * mag


def test_presplit_requires_eligible_shape(ops):
    x = nhwc(rnd((1, 48, 9, 13), 91))
    wt = rnd((64, 48, 1, 1), 92)
    with pytest.raises(RuntimeError):           # Cin % 32 != 0: no LDS-DMA path
        ops.conv_gemm(x, wt, prec="f16x3", split_in=True)


@pytest.mark.parametrize("case", [(2, 45, 80, 1536, 1), (1, 23, 41, 96, 2), (2, 12, 20, 960, 1), (1, 180, 320, 32, 1)])
def test_depthwise_split_output(ops, case):
    """dw3x3 writing its result as a split shadow instead of fp32 (what the projection GEMM stages by DMA)."""
    n, h, w, c, stride = case
    x = rnd((n, c, h, w), 51, 2.0)
    wt = rnd((c, 1, 3, 3), 52, 0.4)
    scale = rnd((c,), 53) * 0.5 + 1.0
    bias = rnd((c,), 54)
    ref = torch.clamp(F.conv2d(x, wt, stride=stride, padding=1, groups=c) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1), 0, 6)
    got = ops.merge_shadow(ops.dw3x3(nhwc(x), wt, scale, bias, stride=stride, split_out=True))
    assert (nchw(got) - ref).abs().max().item() <= 1e-5
    plain = ops.dw3x3(nhwc(x), wt, scale, bias, stride=stride)
    assert (got - plain).abs().max().item() <= 6.0 * 2 ** -20       # two fp16 roundings of values in [0, 6]


def test_bilinear_split_output(ops):
    x = nhwc(rnd((3, 64, 12, 20), 71, 3.0))
    out, sp = ops.bilinear_ac(x, 45, 80, split_out=True)
    assert (ops.merge_shadow(sp) - out).abs().max().item() <= 3.0 * 2 ** -20


# ---- fused inverted-residual block (uavsal_fused_ir): MobileNetV2 features[1..7] in one launch each --------
FUSED_CASES = [
    # n, h, w, cin, hidden, cout, stride, residual
    (2, 20, 36, 32, 32, 16, 1, False),      # features.1 (no expand conv)
    (2, 21, 35, 16, 96, 24, 2, False),      # features.2, odd sizes: partial patches on both edges
    (1, 19, 27, 24, 144, 24, 1, True),      # features.3 (residual)
    (2, 18, 32, 24, 144, 32, 2, False),     # features.4
    (1, 45, 80, 32, 192, 32, 1, True),      # features.5 / .6 at the 360x640 map size
    (3, 9, 13, 32, 192, 64, 2, False),      # features.7
    (1, 5, 3, 16, 96, 24, 2, False),        # smaller than one patch
]


@pytest.mark.parametrize("tile", [1, 2])          # 4x16 patches / 16x16 (stride 1), 8x16 (stride 2)
@pytest.mark.parametrize("case", FUSED_CASES)
def test_fused_inverted_residual(ops, case, tile):
    n, h, w, cin, hid, cout, stride, res = case
    x = rnd((n, cin, h, w), 101, 2.0)
    bn = lambda c, s: (rnd((c,), s) * 0.5 + 1.0, rnd((c,), s + 1))
    w1 = rnd((hid, cin, 1, 1), 102, 1.0 / np.sqrt(cin)) if hid != cin or cin != 32 else None
    wd = rnd((hid, 1, 3, 3), 103, 0.4)
    w2 = rnd((cout, hid, 1, 1), 104, 1.0 / np.sqrt(hid))
    b1, bd, b2 = bn(hid, 105), bn(hid, 107), bn(cout, 109)
    aff = lambda y, b: y * b[0].view(1, -1, 1, 1) + b[1].view(1, -1, 1, 1)
    e = x if w1 is None else torch.clamp(aff(F.conv2d(x, w1), b1), 0, 6)
    dd = torch.clamp(aff(F.conv2d(e, wd, stride=stride, padding=1, groups=hid), bd), 0, 6)
    ref = aff(F.conv2d(dd, w2), b2)
    if res:
        ref = ref + x
    got = ops.fused_ir(nhwc(x), w1, b1, wd, bd, w2, b2, stride=stride, residual=res, tile=tile)
    assert tuple(got.shape) == (n, ref.shape[2], ref.shape[3], cout)
    err = (nchw(got) - ref).abs().max().item()
    assert err <= 2e-5 * max(1.0, ref.abs().max().item()), (case, err)


def test_fused_inverted_residual_rejects_other_shapes(ops):
    x = nhwc(rnd((1, 128, 9, 13), 111))
    with pytest.raises(RuntimeError):
        ops.fused_ir(x, rnd((768, 128, 1, 1), 1), (torch.ones(768), torch.zeros(768)), rnd((768, 1, 3, 3), 2),
                     (torch.ones(768), torch.zeros(768)), rnd((128, 768, 1, 1), 3), (torch.ones(128), torch.zeros(128)))
    with pytest.raises(RuntimeError):          # the mid-channel kernel (csrc/fused_mid.hip) is stride 1 only
        ops.fused_ir(nhwc(rnd((1, 64, 9, 13), 112)), rnd((384, 64, 1, 1), 1), (torch.ones(384), torch.zeros(384)), rnd((384, 1, 3, 3), 2),
                     (torch.ones(384), torch.zeros(384)), rnd((64, 384, 1, 1), 3), (torch.ones(64), torch.zeros(64)), stride=2)


# ---- mid-channel fused block (csrc/fused_mid.hip): features[8..13], the second prior-net block, the temporal sub-block ----
MID_CASES = [   # (n, h, w, cin, hidden, cout): maps that are / are not multiples of the 4 x 8 patch, tiny maps
    (2, 23, 40, 64, 384, 64), (1, 12, 20, 64, 384, 96), (2, 23, 40, 96, 576, 96), (1, 9, 13, 64, 384, 32),
    (3, 4, 8, 64, 384, 64), (1, 5, 9, 96, 576, 96), (1, 1, 1, 64, 384, 64), (1, 45, 80, 64, 384, 32),
]


@pytest.mark.parametrize("case", MID_CASES)
def test_fused_mid_channel_block(ops, case):
    """One launch == pw-expand + BN + ReLU6 -> dw3x3 + BN + ReLU6 -> pw-linear + BN [+ x] (reference model.py:74-103;
    torchvision InvertedResidual), exact fp32, against torch-CPU fp32."""
    n, h, w, cin, hid, cout = case
    x = rnd((n, cin, h, w), 201, 2.0)
    bn = lambda c, s: (rnd((c,), s) * 0.5 + 1.0, rnd((c,), s + 1))
    w1 = rnd((hid, cin, 1, 1), 202, 1.0 / np.sqrt(cin))
    wd = rnd((hid, 1, 3, 3), 203, 0.4)
    w2 = rnd((cout, hid, 1, 1), 204, 1.0 / np.sqrt(hid))
    b1, bd, b2 = bn(hid, 205), bn(hid, 207), bn(cout, 209)
    aff = lambda y, b: y * b[0].view(1, -1, 1, 1) + b[1].view(1, -1, 1, 1)
    e = torch.clamp(aff(F.conv2d(x, w1), b1), 0, 6)
    dd = torch.clamp(aff(F.conv2d(e, wd, padding=1, groups=hid), bd), 0, 6)
    ref = aff(F.conv2d(dd, w2), b2)
    res = cin == cout
    if res:
        ref = ref + x
    got = ops.fused_ir(nhwc(x), w1, b1, wd, bd, w2, b2, stride=1, residual=res)
    assert tuple(got.shape) == (n, h, w, cout)
    err = (nchw(got) - ref).abs().max().item()
    assert err <= 2e-5 * max(1.0, ref.abs().max().item()), (case, err)

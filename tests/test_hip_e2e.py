"""End-to-end parity of the HIP path on the MI355X, through the drop-in `UAVSal` surface.

Checked against (a) the oracle (oracle/uavsal_ref.py, torch-CPU fp32 restatement) on the same
seeded inputs and (b) the committed golden vectors that oracle/make_goldens.py produced by
running the reference's own model.py.  Tolerance on the fp32 saliency map: 1e-3 max-abs
(BASELINE.json north_star); the two SUPPORTED modes -- exact-fp32 MFMA (`f32`, the headline) and the
3 x fp16 split (`f16x3`) -- are gated at 5e-4 on every golden.  `bf16x3` and `bf16` are diagnostics:
they are NOT parity modes (bf16x3 misses 1e-3 at 360x640: 1.15e-3; single-pass bf16, the literal
reading of BASELINE configs[2], misses it by 300x: profiles/r4_precision.md) and the bounds they are
held to here are regression bounds, not claims.
"""
import os

import numpy as np
import pytest
import torch

from iip_uavsal_saliency_amd import synth

pytestmark = pytest.mark.gpu

# The synthetic network amplifies fp32 round-off by ~10^3 (two exact-fp32 implementations that
# only differ in summation order -- this one and oneDNN -- end 1.5e-4 apart on the 360x640 map),
# so the fp32-class modes are held to 5e-4, half the 1e-3 of the north_star.  `bf16x3` (diagnostic, see the
# module docstring) is bounded at 2e-3: a regression bound ABOVE the north_star's tolerance, not a parity claim.
NORTH_STAR_TOL = 1e-3
PARITY_PRECS = ["f32", "f16x3"]            # the modes the package claims; every golden, <= 5e-4
MAP_TOL = {"f32": 5e-4, "f16x3": 5e-4, "bf16x3": 2e-3}
assert all(MAP_TOL[p_] <= NORTH_STAR_TOL for p_ in PARITY_PRECS)
LOGIT_TOL = {"f32": 2e-3, "f16x3": 2e-3, "bf16x3": 1.5e-2}     # logits span about +-12
STATE_TOL = {"f32": 5e-4, "f16x3": 5e-4, "bf16x3": 4e-3}
TAP_REL = {"f32": 1e-4, "f16x3": 1e-4, "bf16x3": 1e-3}        # relative to max|tap|
PRECS = ["f32", "f16x3", "bf16x3"]


def make_inputs(n, H, W, seed=0, t0=0):
    h, w = H // 8, W // 8
    x = torch.from_numpy(synth.normalize_frames(synth.synth_frames_u8(n, H, W, seed, t0)))
    cb = [torch.from_numpy(synth.gauss_priors(n, h, w)), torch.from_numpy(synth.ob_priors(n, h, w, seed=seed))]
    return x, cb


@pytest.fixture(scope="module")
def hip_model():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from iip_uavsal_saliency_amd import UAVSal
    m = UAVSal(time_dims=4)
    synth.load_synth_weights(m, 0)
    return m.cuda().eval()


@pytest.fixture(scope="module")
def oracle():
    from oracle.uavsal_ref import build_oracle
    return build_oracle(time_dims=4, seed=0)


def _run_hip(model, T, prec, x, cb, state=None, taps=None):
    model.time_dims = T
    model.precision = prec
    st = None if state is None else [state.cuda()]
    out, s = model(x.cuda(), [cb[0].cuda(), cb[1].cuda()], st, taps)
    return out.cpu(), s[0].cpu()


@pytest.mark.parametrize("prec", PRECS)
def test_forward_vs_oracle_with_taps(hip_model, oracle, prec):
    x, cb = make_inputs(4, 96, 160)
    oracle.time_dims = 4
    rt = {}
    ro, rs = oracle(x, cb, None, rt)
    ht = {}
    ho, hs = _run_hip(hip_model, 4, prec, x, cb, None, ht)
    for k in ("c3", "c4", "c5", "sfnet", "st0", "st1", "fust_in_cb", "prefuse", "rnn"):
        err = (ht[k].cpu() - rt[k]).abs().max().item()
        assert err <= TAP_REL[prec] * rt[k].abs().max().item(), (k, prec, err)
    assert (ht["logits"].cpu() - rt["logits"]).abs().max().item() <= LOGIT_TOL[prec]
    assert (ho - ro).abs().max().item() <= MAP_TOL[prec]
    assert (hs - rs[0]).abs().max().item() <= STATE_TOL[prec]


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("name", ["e2e_96x160_T4", "e2e_96x160_B4T5", "e2e_96x160_T4_two_calls",
                                  "e2e_72x104_T3", "e2e_288x512_T8", "e2e_360x640_T8"])
def test_forward_vs_reference_golden(hip_model, golden_dir, name, prec):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    H, W, T, B = int(g["H"]), int(g["W"]), int(g["T"]), int(g["B"])
    n = B * T
    state = None
    for c in range(int(g["calls"])):
        x, cb = make_inputs(n, H, W, int(g["seed"]), t0=c * n)
        out, st = _run_hip(hip_model, T, prec, x, cb, state)
        state = st
        sfx = "" if c == 0 else f"_call{c}"
        err = np.abs(out.numpy() - g["out" + sfx]).max()
        assert err <= MAP_TOL[prec], (name, prec, c, err)
        serr = np.abs(st.contiguous().view(-1).numpy()[::int(g["state_stride"])] - g["state" + sfx]).max()
        assert serr <= STATE_TOL[prec], (name, prec, c, serr)
    # every stream-K launch must leave its workspace zeroed (nothing published and never collected)
    assert all(e.streamk_clean() for e in hip_model._engines.values())


@pytest.mark.parametrize("prec", PARITY_PRECS)
@pytest.mark.parametrize("name", ["e2e_96x160_T4_bias101", "e2e_96x160_B2T4_bias010", "e2e_96x160_T4_bias000_two_calls",
                                  "e2e_96x160_B2T4_bias001"])
def test_every_bias_type_vs_reference_golden(golden_dir, name, prec):
    """Constructor values other than the Demo default (reference model.py:281-324, 346-365): a disabled prior has no net (and
    its `cb` entry is never read), the concat keeps the order gauss | observed | context, `fucb_layer` takes 64 channels per
    enabled prior, and with no prior at all the recurrence reads `fust_layer`'s output.  Against the reference's own outputs;
    the oracle is run beside it for the taps."""
    from iip_uavsal_saliency_amd import UAVSal
    from oracle.uavsal_ref import build_oracle
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    H, W, T, B = int(g["H"]), int(g["W"]), int(g["T"]), int(g["B"])
    bias = [int(b) for b in g["bias_type"]]
    m = UAVSal(time_dims=T, bias_type=bias)
    synth.load_synth_weights(m, int(g["seed"]))
    m = m.cuda().eval()
    assert m.num_cb == sum(bias) and hasattr(m, "fucb_layer") == bool(sum(bias))
    ora = build_oracle(time_dims=T, seed=int(g["seed"]), bias_type=bias)
    assert list(m.state_dict().keys()) == list(ora.state_dict().keys())
    n = B * T
    state = None
    for c in range(int(g["calls"])):
        x, cb = make_inputs(n, H, W, int(g["seed"]), t0=c * n)
        taps = {}
        # entries of disabled priors are never touched: hand over None in their place (the reference only indexes what it uses)
        cbd = [cb[0] if bias[0] else None, cb[1] if bias[1] else None]
        if not bias[1]:
            cbd = cbd[:1] if bias[0] else []
        m.precision = prec
        out, st = m(x.cuda(), [None if t is None else t.cuda() for t in cbd],
                    None if state is None else [state.cuda()], taps)
        out, st = out.cpu(), st[0].cpu()
        state = st
        sfx = "" if c == 0 else f"_call{c}"
        err = np.abs(out.numpy() - g["out" + sfx]).max()
        lerr = np.abs(taps["logits"].cpu().numpy() - g["logits" + sfx]).max()
        serr = np.abs(st.contiguous().view(-1).numpy()[::int(g["state_stride"])] - g["state" + sfx]).max()
        print("%s %s call %d: map %.3e logits %.3e state %.3e" % (name, prec, c, err, lerr, serr))
        assert err <= MAP_TOL[prec], (name, prec, c, err)
        assert lerr <= 2 * LOGIT_TOL[prec], (name, prec, c, lerr)
        assert serr <= STATE_TOL[prec], (name, prec, c, serr)
        if c == 0:
            for k in ("sfnet", "st1", "fust_in_cb", "prefuse", "rnn"):
                if "tap_" + k not in g:
                    assert not any(bias) and k in ("fust_in_cb", "prefuse")
                    continue
                tap = taps[k].cpu().contiguous().view(-1).numpy()[::int(g["tap_stride"])]
                assert np.abs(tap - g["tap_" + k]).max() <= TAP_REL[prec] * max(1.0, np.abs(g["tap_" + k]).max()), (name, k)
    assert all(e.streamk_clean() for e in m._engines.values())


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
def test_demo_default_call_at_full_size_vs_reference_golden(hip_model, golden_dir, prec):
    """Demo_Test.py:110-125 as it really runs: batch_size=4, time_dims=5 -> ONE forward of 20 frames at 360x640
    (context prior tiled per model.py:357-361, temporal differences across the chunk borders per model.py:194-198),
    then a second call fed with the returned state (Demo_Test.py:86) -- against the reference's own outputs."""
    g = np.load(os.path.join(golden_dir, "e2e_360x640_B4T5_two_calls.npz"))
    H, W, T, B = int(g["H"]), int(g["W"]), int(g["T"]), int(g["B"])
    assert (H, W, T, B, int(g["calls"])) == (360, 640, 5, 4, 2)
    n = B * T
    state = None
    for c in range(2):
        x, cb = make_inputs(n, H, W, int(g["seed"]), t0=c * n)
        taps = {}
        out, st = _run_hip(hip_model, T, prec, x, cb, state, taps)
        state = st
        sfx = "" if c == 0 else f"_call{c}"
        err = np.abs(out.numpy() - g["out" + sfx]).max()
        lerr = np.abs(taps["logits"].cpu().numpy() - g["logits" + sfx]).max()
        serr = np.abs(st.contiguous().view(-1).numpy()[::int(g["state_stride"])] - g["state" + sfx]).max()
        print("demo default %s call %d: map %.3e logits %.3e state %.3e" % (prec, c, err, lerr, serr))
        assert err <= MAP_TOL[prec], (prec, c, err)
        assert lerr <= 2 * LOGIT_TOL[prec], (prec, c, lerr)
        assert serr <= STATE_TOL[prec], (prec, c, serr)
        if c == 0:      # the tiled context reaches the map through fucb_layer: frame k carries chunk k mod B
            tap = taps["fust_in_cb"].cpu().contiguous().view(-1).numpy()[::int(g["tap_stride"])]
            assert np.abs(tap - g["tap_fust_in_cb"]).max() <= TAP_REL[prec] * max(1.0, np.abs(g["tap_fust_in_cb"]).max())
    assert all(e.streamk_clean() for e in hip_model._engines.values())


def test_winograd_and_direct_3x3_agree_end_to_end(hip_model, golden_dir):
    """Exact-fp32 mode with the dense 3x3 convs as Winograd F(2x2, 3x3) (default) and as implicit GEMMs: both within the
    parity bound of the reference's own output at 360x640, and within 2e-4 of each other (oracle experiment: 5e-5)."""
    g = np.load(os.path.join(golden_dir, "e2e_360x640_T8.npz"))
    x, cb = make_inputs(8, 360, 640, int(g["seed"]))
    outs = {}
    try:
        for wino in (True, False):
            hip_model.winograd = wino
            outs[wino], _ = _run_hip(hip_model, 8, "f32", x, cb)
            err = np.abs(outs[wino].numpy() - g["out"]).max()
            print("winograd=%s: map max-abs vs reference golden %.3e" % (wino, err))
            assert err <= MAP_TOL["f32"], (wino, err)
    finally:
        hip_model.winograd = True
    assert (outs[True] - outs[False]).abs().max().item() <= 2e-4


def test_virtual_concat_winograd_input_end_to_end(hip_model, golden_dir):
    """The opt-in form of the SRF-Net head (engine.WINO_SEG, measured not faster and therefore off): `conv_last`'s Winograd input
    transform reads cat[interpolate(x5), interpolate(x4), lv3] itself (uavsal_wino_desc.n_seg, reference model.py:151-156) -- no
    resize launches, no concat buffer.  Same maps as the default plan up to FMA contraction in the resize, inside the parity bound
    of the reference's output; arena in NaN-poisoning debug mode, so the shorter live ranges are checked too."""
    from iip_uavsal_saliency_amd import engine as E
    g = np.load(os.path.join(golden_dir, "e2e_360x640_T8.npz"))
    x, cb = make_inputs(8, 360, 640, int(g["seed"]))
    outs, saved = {}, E.WINO_SEG
    try:
        for mode in (0, 1, 2):
            E.WINO_SEG = mode
            hip_model.invalidate_engines()
            hip_model.arena_debug = mode != 0
            outs[mode], _ = _run_hip(hip_model, 8, "f32", x, cb)
            eng = list(hip_model._engines.values())[-1]
            names = [o["name"] for o in eng.ops_meta]
            assert ("up_c5" in names) == (mode == 0) and ("srf_cat" in eng.named) == (mode == 0)
            err = np.abs(outs[mode].numpy() - g["out"]).max()
            assert err <= MAP_TOL["f32"], (mode, err)
    finally:
        E.WINO_SEG = saved
        hip_model.arena_debug = False
        hip_model.invalidate_engines()
    assert (outs[1] - outs[0]).abs().max().item() <= 5e-5 and torch.equal(outs[1], outs[2])


def test_winograd_modes_agree_at_eight_clips(hip_model, golden_dir):
    """From four clips up the default exact-fp32 plan switches the recurrence steps to F(4x4): default,
    strict F(2x2) everywhere (`model.winograd_r = 2`) and direct 3x3 convs, on BASELINE configs[2]'s 8 clips x 8 frames --
    each inside the parity bound of the reference's own maps, and within 3e-4 of each other."""
    g = np.load(os.path.join(golden_dir, "clips_360x640_C8_T8.npz"))
    H, W, T, C, seed, ms = int(g["H"]), int(g["W"]), int(g["T"]), int(g["C"]), int(g["seed"]), int(g["map_stride"])
    x, cb = make_clips(C, T, H, W, seed)
    x, cb = x.cuda(), [cb[0].cuda(), cb[1].cuda()]
    hip_model.precision = "f32"
    outs = {}
    try:
        for mode, (wino, r) in {"default": (True, None), "strict F(2x2)": (True, 2), "direct": (False, None)}.items():
            hip_model.winograd, hip_model.winograd_r = wino, r
            out, _ = hip_model.forward_clips(x, cb, None)
            outs[mode] = out.cpu()
            eng = next(reversed(hip_model._engines.values()))
            if wino:
                assert eng.winograd_step_r == (2 if r == 2 else 0) and eng.winograd_r == (2 if r == 2 else 4)
            err = np.abs(outs[mode].contiguous().view(-1).numpy()[::ms] - g["out"]).max()
            print("%s: map max-abs vs reference golden %.3e" % (mode, err))
            assert err <= MAP_TOL["f32"], (mode, err)
    finally:
        hip_model.winograd, hip_model.winograd_r = True, None
    for a in outs:
        for b in outs:
            assert (outs[a] - outs[b]).abs().max().item() <= 3e-4, (a, b)


def test_default_fp32_plan_composition(hip_model):
    """What DESIGN.md section 10 says about the default exact-fp32 plan at the benchmark shape (360x640, 1 clip x 8 frames),
    pinned: no stream-K launch is left, the dense 3x3 convs are Winograd triples, the three dilated ASPP projections are one
    grouped launch, the 384-hidden blocks at 45x80 run depthwise + projection fused, features[1..7] are one launch each, and
    the small-map blocks (patch grid > 15 % outside the map) keep separate depthwise / projection launches."""
    hip_model.precision, hip_model.time_dims = "f32", 8          # (whatever an earlier test left on the shared model)
    eng = hip_model._engine(torch.device("cuda", torch.cuda.current_device()), 1, 8, 360, 640, "tile")
    meta = {m["name"]: m for m in eng.ops_meta}
    assert all(m.get("streamk", 0) == 0 for m in eng.ops_meta)
    for nm in ("conv_last", "twa.wx", "twa.step0", "twa.step7"):
        assert nm + ".xin" in meta and nm + ".xout" in meta and meta[nm]["tile"] == 8, nm
    assert "aspp.pl" in meta and meta["aspp.pl"]["tile"] == 11 and meta["aspp.pl"]["Nc"] == 768 and meta["aspp.pl"]["K"] == 1920
    assert not any(n in meta for n in ("aspp2.pl", "aspp3.pl", "aspp4.pl"))
    assert "aspp.dw" in meta and meta["aspp.dw"]["dil"] == (6, 12, 18) and not any(n in meta for n in ("aspp2.dw", "aspp3.dw", "aspp4.dw"))
    for nm in ("st0.sub", "st1.sub", "gauss.1", "ob.1", "st0.sp", "fust", "fucbst"):
        assert nm + ".dwpl" in meta and meta[nm + ".dwpl"]["dwproj"] != 0, nm
    # the decoder's 1536 -> 1 projection: a dot product per pixel inside the depthwise launch, not a GEMM + reduce launch
    assert meta["conv_out_st.dwpl"]["kind"] == "dw_dot" and meta["conv_out_st.dwpl"]["kernel"].startswith("dw3x3_dot_kernel")
    for i in range(1, 8):
        assert meta["features.%d" % i]["kind"] == "fused_ir" and meta["features.%d" % i]["kernel"].startswith("fused_ir_kernel")
    # round 4: the stride-1 blocks of the 23x40 map (features.8-13: 240 workgroups of the mid-channel kernel) are one launch each
    for i in range(8, 14):
        assert meta["features.%d" % i]["kind"] == "fused_ir" and meta["features.%d" % i]["kernel"].startswith("fused_mid_kernel")
    for i in (14, 15, 17):       # stride 2 / the 12x20 map: three launches
        assert "features.%d.dw" % i in meta and "features.%d.pl" % i in meta and "features.%d.dwpl" % i not in meta
    # ... and the same block shapes at 45x80 (960 workgroups: several rounds of the chip) keep their GEMM + fused dw/projection launches
    for nm in ("gauss.1", "st0.sub"):
        assert nm + ".pw" in meta and nm + ".dwpl" in meta
    assert sum(1 for m in eng.ops_meta if m["kind"] != "sync") <= 118


def make_clips(C, T, H, W, seed=0, t0=0):
    """Clip c is seeded with seed + c (oracle/make_goldens.py clip_inputs, bench.py make_clips)."""
    h, w = H // 8, W // 8
    xs, g, o = [], [], []
    for c in range(C):
        xs.append(torch.from_numpy(synth.normalize_frames(synth.synth_frames_u8(T, H, W, seed + c, t0))))
        g.append(torch.from_numpy(synth.gauss_priors(T, h, w)))
        o.append(torch.from_numpy(synth.ob_priors(T, h, w, seed=seed + c)))
    return torch.stack(xs), [torch.stack(g), torch.stack(o)]


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
@pytest.mark.parametrize("name", ["clips_360x640_C8_T8", "clips_720x1280_C4_T16_two_calls"])
def test_forward_clips_vs_reference_golden(hip_model, golden_dir, name, prec):
    """BASELINE configs[2] (360x640, 8 clips x 8 frames; also one GPU's share of configs[3]) and configs[4]
    (720x1280, 4 clips x 16 frames, two successive calls with the state carried) against what the
    REFERENCE's model.py produced for the same clips as C independent calls (model.py:341-375, state carry
    Demo_Test.py:75-86).  `f16x3` is the split-16-bit MFMA mode that stands in for configs[2]'s "bf16"
    (plain bf16 misses 1e-3 by 250x, see test_bf16_single_pass_error_is_reported)."""
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    H, W, T, C, seed = int(g["H"]), int(g["W"]), int(g["T"]), int(g["C"]), int(g["seed"])
    ms, ss = int(g["map_stride"]), int(g["state_stride"])
    hip_model.precision = prec
    states = None
    for call in range(int(g["calls"])):
        x, cb = make_clips(C, T, H, W, seed, t0=call * T)
        out, states = hip_model.forward_clips(x.cuda(), [cb[0].cuda(), cb[1].cuda()], states)
        del x, cb
        sfx = "" if call == 0 else f"_call{call}"
        o = out.cpu()
        assert tuple(o.shape) == (C, T, 1, H // 8, W // 8)
        err = np.abs(o.contiguous().view(-1).numpy()[::ms] - g["out" + sfx]).max()
        err0 = np.abs(o[0].numpy() - g["out_clip0" + sfx]).max()
        serr = np.abs(states.cpu().contiguous().view(-1).numpy()[::ss] - g["state" + sfx]).max()
        print("%s %s call %d: map %.3e (clip 0 full %.3e) state %.3e" % (name, prec, call, err, err0, serr))
        assert err <= MAP_TOL[prec] and err0 <= MAP_TOL[prec], (name, prec, call, err, err0)
        assert serr <= STATE_TOL[prec], (name, prec, call, serr)
        # float64 checksums of the whole tensors (the strided samples skip most state values)
        n_out, n_st = o.numel(), states.numel()
        assert abs(o.double().sum().item() - float(g["out_sum" + sfx])) <= MAP_TOL[prec] * n_out * 0.05
        assert abs(states.double().sum().item() - float(g["state_sum" + sfx])) <= STATE_TOL[prec] * n_st * 0.05
    assert all(e.streamk_clean() for e in hip_model._engines.values())
    hip_model.invalidate_engines()          # the 720p plan holds ~20 GB of activations: release it


@pytest.mark.parametrize("prec", PARITY_PRECS)
@pytest.mark.parametrize("shape", [(1, 4, 96, 160), (2, 4, 96, 160), (4, 3, 72, 104), (1, 8, 360, 640)])
def test_arena_is_bit_identical_and_never_reads_a_released_range(hip_model, prec, shape):
    """The activation arena (engine.py: one pool per plan, placed by liveness) against one allocation per activation: the
    same launches on other addresses, so maps and states are identical BIT FOR BIT -- and again with `arena_debug`, where
    every range is NaN-filled right after its last declared use (a use after release, or a buffer placed over a live one,
    cannot stay finite); taps included (they pin their buffers to the end of the plan); persistent state; two calls."""
    C, T, H, W = shape
    calls = [make_clips(C, T, H, W, 0, t0=k * T) for k in range(2)]
    calls = [(x.cuda(), [cb[0].cuda(), cb[1].cuda()]) for x, cb in calls]
    hip_model.precision = prec
    res = {}
    try:
        for mode in ("unshared", "arena", "debug", "debug+persistent"):
            hip_model.arena = mode != "unshared"
            hip_model.arena_debug = mode.startswith("debug")
            hip_model.persistent_state = mode.endswith("persistent")
            st, outs = None, []
            for x, cb in calls:
                taps = {} if mode == "debug" else None
                o, st = hip_model.forward_clips(x, cb, st.detach() if st is not None else None, taps)
                outs.append((o.clone(), st.clone()))
                if taps is not None:
                    assert all(bool(torch.isfinite(t).all().item()) for t in taps.values())
            res[mode] = outs
            eng = list(hip_model._engines.values())[-1]
            assert eng.use_arena == (mode != "unshared")
            if mode.startswith("debug"):
                assert any(o["kind"] == "poison" for o in eng.ops_meta)
                # (with taps the buffers the test reads back are pinned to the end of the plan; the rest is poisoned)
            if mode == "arena":
                st_ = eng.arena_stats
                print("arena %s %s: %.1f MB (live bound %.1f, unshared %.1f)" % (shape, prec, st_["arena_mb"], st_["live_bound_mb"], st_["unshared_mb"]))
                assert st_["arena_mb"] < 0.45 * st_["unshared_mb"]
        for mode in ("arena", "debug", "debug+persistent"):
            for k in range(2):
                assert bool(torch.isfinite(res[mode][k][0]).all().item()), (mode, k)
                assert torch.equal(res[mode][k][0], res["unshared"][k][0]), (mode, k, (res[mode][k][0] - res["unshared"][k][0]).abs().max().item())
                assert torch.equal(res[mode][k][1], res["unshared"][k][1]), (mode, k)
    finally:
        hip_model.arena, hip_model.arena_debug, hip_model.persistent_state = True, False, False
        if H >= 360:
            hip_model.invalidate_engines()


@pytest.mark.parametrize("prec", PARITY_PRECS)
def test_arena_on_the_reference_surface_with_taps_and_lstm(hip_model, prec):
    """`forward()` -- the reference's own surface: one sequence of B x time_dims frames, context prior tiled, taps read back after
    the run -- and the ConvLSTM variant (cell-state history buffer), each as one allocation per activation vs the NaN-poisoned arena:
    maps, states and every tap bit-identical."""
    from iip_uavsal_saliency_amd import UAVSAL_LSTM
    x, cb = make_inputs(20, 96, 160)
    xd, cbd = x.cuda(), [cb[0].cuda(), cb[1].cuda()]
    lstm = UAVSAL_LSTM(time_dims=5)
    synth.load_synth_weights(lstm, 0)
    lstm = lstm.cuda().eval()
    for m in (hip_model, lstm):
        m.time_dims, m.precision = 5, prec
        got = {}
        try:
            for mode in ("unshared", "debug"):
                m.arena, m.arena_debug = mode != "unshared", mode == "debug"
                taps = {}
                o1, s1 = m(xd, cbd, None, taps)
                o2, s2 = m(xd, cbd, [tuple(s1)] if m.rnn_type == "lstm" else s1, None)           # second call, carried state, no taps
                got[mode] = ([o1.clone(), o2.clone()] + [t.clone() for t in s1] + [t.clone() for t in s2], {k: v.clone() for k, v in taps.items()})
                assert list(m._engines.values())[-1].use_arena == (mode != "unshared")
        finally:
            m.arena, m.arena_debug = True, False
        for a, b in zip(got["unshared"][0], got["debug"][0]):
            assert torch.equal(a, b) and bool(torch.isfinite(b).all().item())
        assert sorted(got["unshared"][1]) == sorted(got["debug"][1]) and len(got["debug"][1]) >= 9
        for k in got["unshared"][1]:
            assert torch.equal(got["unshared"][1][k], got["debug"][1][k]), k


def test_persistent_state_equals_refed_state(hip_model):
    """Opt-in persistent-state mode (BASELINE configs[4], SURVEY.md 8(b) Ownership): the state stays in the
    engine's NHWC buffer between calls == re-feeding the returned state, bit for bit; a foreign tensor is
    loaded, None resets, and the default mode still never aliases."""
    C, T, H, W = 2, 4, 96, 160
    calls = [make_clips(C, T, H, W, 0, t0=k * T) for k in range(3)]
    calls = [(x.cuda(), [cb[0].cuda(), cb[1].cuda()]) for x, cb in calls]
    hip_model.precision = "f32"
    hip_model.persistent_state = False
    ref, st = [], None
    for x, cb in calls:
        o, st = hip_model.forward_clips(x, cb, st)
        ref.append((o.clone(), st.clone()))
    hip_model.persistent_state = True
    try:
        st = None
        for k, (x, cb) in enumerate(calls):
            o, st = hip_model.forward_clips(x, cb, st.detach() if st is not None else None)
            assert torch.equal(o, ref[k][0]) and torch.equal(st, ref[k][1]), k
        eng = [e for e in hip_model._engines.values() if e.persistent][-1]
        assert st.data_ptr() == eng.h_view.data_ptr()                     # a view, not a copy
        # a foreign state tensor (here: call 0's, from the non-persistent run) is loaded into the buffer
        o, st = hip_model.forward_clips(calls[1][0], calls[1][1], ref[0][1])
        assert torch.equal(o, ref[1][0]) and torch.equal(st, ref[1][1])
        o, st = hip_model.forward_clips(calls[0][0], calls[0][1], None)   # None resets to zeros
        assert torch.equal(o, ref[0][0])
        # the reference surface (forward, one sequence): persistent == re-fed, bit for bit
        hip_model.time_dims = T
        x1, cb1 = calls[0][0][0], [calls[0][1][0][0], calls[0][1][1][0]]
        x2, cb2 = calls[1][0][0], [calls[1][1][0][0], calls[1][1][1][0]]
        hip_model.persistent_state = False
        r1, rs1 = hip_model(x1, cb1, None)
        r2, rs2 = hip_model(x2, cb2, [rs1[0].detach()])
        hip_model.persistent_state = True
        o1, s1 = hip_model(x1, cb1, None)
        assert torch.equal(o1, r1) and torch.equal(s1[0], rs1[0])
        o2, s2 = hip_model(x2, cb2, [s1[0].detach()])
        assert torch.equal(o2, r2) and torch.equal(s2[0], rs2[0])
    finally:
        hip_model.persistent_state = False


def test_lost_streamk_piece_raises_and_poisons(hip_model):
    """A stream-K piece that is never published must not produce a map (SURVEY.md 8(b) Errors: never silent):
    with the test hook every producer withholds its flag, the owners' bounded wait gives up, the guard op
    overwrites map and state with NaN, and forward raises -- in the same call with sync_errors (default),
    at check_errors() / the next call without."""
    # (at the benchmark size: the 3x3 convs of the head run stream-K there -- 450 tiles of 128 x 128; on small maps the
    # launches that used to are now K-split instances that hand nothing over between resident workgroups)
    T, H, W = 8, 360, 640
    x, cb = make_inputs(T, H, W)
    args = (x.cuda(), [cb[0].cuda(), cb[1].cuda()], None)
    hip_model.time_dims, hip_model.precision = T, "f32"
    hip_model.winograd = False                # the direct 3x3 convs are the stream-K launches (Winograd plans have none)
    good, _ = hip_model(*args)
    hip_model._sk_debug = (2000, -1)          # (poll limit, withhold every published flag)
    try:
        eng_sk = sum(m.get("streamk", 0) > 0 for m in hip_model._engine(x.cuda().device, 1, T, H, W, "tile").ops_meta)
        assert eng_sk > 0, "no stream-K launch in this plan: the test would prove nothing"
        with pytest.raises(RuntimeError, match="stream-K"):
            hip_model(*args)
        hip_model.sync_errors = False
        out, st = hip_model(*args)            # asynchronous: the poisoned result is returned ...
        torch.cuda.synchronize()
        assert torch.isnan(out).all() and torch.isnan(st[0]).all()
        with pytest.raises(RuntimeError, match="stream-K"):
            hip_model.check_errors()          # ... and the error is reported at the next check
        # forward_clips (throughput surface) is asynchronous by default: poisoned result, error at the next check
        hip_model.sync_errors = None
        xc = args[0].view(1, T, 3, H, W)
        oc, sc = hip_model.forward_clips(xc, [args[1][0].view(1, T, 8, H // 8, W // 8), args[1][1].view(1, T, 20, H // 8, W // 8)], None)
        torch.cuda.synchronize()
        assert torch.isnan(oc).all() and torch.isnan(sc).all()
        with pytest.raises(RuntimeError, match="stream-K"):
            hip_model.check_errors()
    finally:
        hip_model._sk_debug = (0, 0)
        hip_model.sync_errors = None
    try:
        again, _ = hip_model(*args)               # workspaces were re-zeroed: the healthy plan still works
        assert torch.equal(again, good)
        assert all(e.streamk_clean() for e in hip_model._engines.values())
    finally:
        hip_model.winograd = True


def test_bf16_single_pass_error_is_reported(hip_model, oracle):
    """Plain bf16 MFMA inputs do not meet 1e-3 in general; keep it bounded and visible."""
    x, cb = make_inputs(4, 96, 160)
    oracle.time_dims = 4
    ro, _ = oracle(x, cb, None)
    ho, _ = _run_hip(hip_model, 4, "bf16", x, cb)
    err = (ho - ro).abs().max().item()
    print("bf16 single-pass max-abs on the map: %.3e" % err)
    assert err <= 0.5


def test_diagnostic_precisions_on_the_eight_clip_golden(hip_model, golden_dir):
    """`bf16x3` and single-pass `bf16` on BASELINE configs[2]'s shape (8 clips x 8 frames, 360x640) against the
    reference's own maps: reported, bounded (regression bounds), and explicitly NOT inside the north_star's 1e-3 --
    configs[2]'s literal "bf16" is not delivered; `f16x3` (test_forward_clips_vs_reference_golden) is what stands in."""
    g = np.load(os.path.join(golden_dir, "clips_360x640_C8_T8.npz"))
    H, W, T, C, seed, ms = int(g["H"]), int(g["W"]), int(g["T"]), int(g["C"]), int(g["seed"]), int(g["map_stride"])
    x, cb = make_clips(C, T, H, W, seed)
    x, cb = x.cuda(), [cb[0].cuda(), cb[1].cuda()]
    errs = {}
    try:
        for prec in ("bf16x3", "bf16"):
            hip_model.precision = prec
            out, _ = hip_model.forward_clips(x, cb, None)
            errs[prec] = float(np.abs(out.cpu().contiguous().view(-1).numpy()[::ms] - g["out"]).max())
            print("%s on clips_360x640_C8_T8: map max-abs vs the reference %.3e" % (prec, errs[prec]))
    finally:
        hip_model.precision = "f32"
    assert errs["bf16x3"] <= MAP_TOL["bf16x3"]          # regression bound (2e-3), above the 1e-3 tolerance
    assert NORTH_STAR_TOL < errs["bf16"] <= 0.5          # single-pass bf16 does not meet 1e-3: keep that visible


@pytest.mark.parametrize("prec", PRECS)
def test_forward_clips_equals_independent_reference_calls(hip_model, oracle, prec):
    C, T, H, W = 3, 3, 72, 104
    x, cb = make_inputs(C * T, H, W)
    h, w = H // 8, W // 8
    xc = x.view(C, T, 3, H, W)
    cbc = [cb[0].view(C, T, 8, h, w), cb[1].view(C, T, 20, h, w)]
    g = torch.Generator().manual_seed(5)
    st0 = torch.rand((C, 256, h, w), generator=g)
    ro, rs = oracle.forward_clips(xc, cbc, st0)
    hip_model.precision = prec
    ho, hs = hip_model.forward_clips(xc.cuda(), [cbc[0].cuda(), cbc[1].cuda()], st0.cuda())
    assert (ho.cpu() - ro).abs().max().item() <= MAP_TOL[prec]
    assert (hs.cpu() - rs).abs().max().item() <= STATE_TOL[prec]


def test_state_is_not_aliased_between_calls(hip_model):
    """The caller re-feeds `[out_state[0].detach()]` (Demo_Test.py:86): a returned state must
    stay valid after the next call."""
    x, cb = make_inputs(4, 96, 160)
    o1, s1 = _run_hip(hip_model, 4, "f32", x, cb)
    keep = s1.clone()
    x2, cb2 = make_inputs(4, 96, 160, t0=4)
    _run_hip(hip_model, 4, "f32", x2, cb2, s1)
    assert torch.equal(s1, keep)


def test_uint8_frames_match_float_frames(hip_model):
    u8 = synth.synth_frames_u8(4, 96, 160)
    x, cb = make_inputs(4, 96, 160)
    hip_model.time_dims, hip_model.precision = 4, "f32"
    a, _ = hip_model(x.cuda(), [cb[0].cuda(), cb[1].cuda()], None)
    b, _ = hip_model(torch.from_numpy(u8).cuda(), [cb[0].cuda(), cb[1].cuda()], None)
    assert (a - b).abs().max().item() <= 1e-4


def test_frame_invariant_priors_run_the_prior_nets_once(hip_model, oracle):
    """The reference's caller hands the SAME prior maps to every frame (np.repeat, utils_data.py:466-467, 601-602).  Given as a
    zero-stride view (priors.get_bias's default) the plan runs the two prior nets on one frame and broadcasts their output; given
    materialised -- or with model.dedupe_priors off -- every frame goes through them.  Same maps either way (the one-frame launches
    may pick other tiles: summation order), both within the parity bound of the oracle."""
    n, H, W = 8, 96, 160
    x, cb = make_inputs(n, H, W)
    assert all(torch.equal(c[0], c[k]) for c in cb for k in range(n))            # synth priors: one map set per call, like get_bias
    hip_model.time_dims, hip_model.precision = 4, "f32"
    xc = x.cuda()
    mat = [c.cuda() for c in cb]
    view = [c[:1].cuda().expand(n, -1, -1, -1) for c in cb]
    out_g, st_g = hip_model(xc, mat, None)
    out_s, st_s = hip_model(xc, view, None)
    engs = {e.static_priors: e for e in hip_model._engines.values() if e.N == n and e.H == H and e.prec_name == "f32"}
    assert set(engs) == {False, True}
    names = lambda e: [m["name"] for m in e.ops_meta]
    assert "gauss.bcast" in names(engs[True]) and "ob.bcast" in names(engs[True]) and "gauss.bcast" not in names(engs[False])
    assert len(names(engs[True])) == len(names(engs[False])) + 2
    assert (out_s - out_g).abs().max().item() <= 1e-4 and (st_s[0] - st_g[0]).abs().max().item() <= 1e-4
    oracle.time_dims = 4
    ro, rs = oracle(x, cb, None)
    assert (out_s.cpu() - ro).abs().max().item() <= MAP_TOL["f32"] and (st_s[0].cpu() - rs[0]).abs().max().item() <= STATE_TOL["f32"]
    hip_model.dedupe_priors = False
    try:
        out_off, _ = hip_model(xc, view, None)                                   # the view through the general plan: bit for bit
    finally:
        hip_model.dedupe_priors = True
    assert torch.equal(out_off, out_g)
    # forward_clips: [C, T, ., h, w] broadcast over clips AND frames
    x5 = xc.view(2, 4, 3, H, W)
    v5 = [c[:1].cuda()[None].expand(2, 4, -1, -1, -1) for c in cb]
    m5 = [c.cuda().view(2, 4, *c.shape[1:]) for c in cb]
    o5s, s5s = hip_model.forward_clips(x5, v5, None)
    o5g, s5g = hip_model.forward_clips(x5, m5, None)
    hip_model.check_errors()
    assert any(e.static_priors and e.n_seq == 2 for e in hip_model._engines.values())
    assert (o5s - o5g).abs().max().item() <= 1e-4 and (s5s - s5g).abs().max().item() <= 1e-4


def test_graph_replay_matches_launch_loop(hip_model):
    x, cb = make_inputs(4, 96, 160)
    a, sa = _run_hip(hip_model, 4, "f16x3", x, cb)
    hip_model.use_graph = True
    try:
        b, sb = _run_hip(hip_model, 4, "f16x3", x, cb)
        b2, _ = _run_hip(hip_model, 4, "f16x3", x, cb)
    finally:
        hip_model.use_graph = False
    assert torch.equal(a, b) and torch.equal(sa, sb) and torch.equal(a, b2)


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
def test_fused_depthwise_projection_everywhere_matches_three_launches(hip_model, prec):
    """`fuse_dw=True` puts EVERY dilation-1 block's depthwise inside its projection launch (the default only does so
    for the big blocks: at this size none), `False` none: the maps must agree to rounding."""
    x, cb = make_inputs(4, 96, 160)
    try:
        hip_model.fuse_dw = False
        a, sa = _run_hip(hip_model, 4, prec, x, cb)
        hip_model.fuse_dw = True
        b, sb = _run_hip(hip_model, 4, prec, x, cb)
    finally:
        hip_model.fuse_dw = None
    tol = 2e-4 if prec == "f32" else 3e-4     # (fp32 summation orders differ in ~20 layers: ~5e-5 on the map)
    assert (a - b).abs().max().item() <= tol and (sa - sb).abs().max().item() <= tol * 5


def test_error_behaviour(hip_model):
    x, cb = make_inputs(4, 96, 160)
    hip_model.time_dims = 4
    with pytest.raises(RuntimeError):     # a single frame: the reference raises in teConv_sub
        hip_model(x[:1].cuda(), [cb[0][:1].cuda(), cb[1][:1].cuda()], None)
    with pytest.raises(RuntimeError):     # not a multiple of time_dims: x.view(B, T, ...) fails
        hip_model(x[:3].cuda(), [cb[0][:3].cuda(), cb[1][:3].cuda()], None)
    with pytest.raises(RuntimeError):     # wrong prior shape
        hip_model(x.cuda(), [cb[0][:, :4].cuda(), cb[1].cuda()], None)
    with pytest.raises(RuntimeError):     # CPU tensors: no fallback
        hip_model(x, cb, None)


def test_postprocess_matches_numpy_restatement():
    """Device resize + crop + /max*255 + rint vs oracle/post_ref.py (cv2 itself is not installed:
    this row is 'parity unpinned', see the oracle's header).  uint8 may differ by one count where
    the fp32 value sits on a .5 boundary."""
    from iip_uavsal_saliency_amd import ops
    from oracle import post_ref
    g = torch.Generator().manual_seed(3)
    for (h, w, H, W) in [(45, 80, 360, 640), (45, 80, 300, 640), (45, 80, 720, 1000), (36, 64, 288, 512)]:
        maps = torch.rand((3, 1, h, w), generator=g) * 0.9 + 0.05
        got = ops.postprocess_predictions(maps.cuda(), H, W).cpu().numpy()
        for i in range(3):
            ref = post_ref.to_uint8(post_ref.postprocess_predictions(maps[i, 0].numpy(), H, W))
            diff = np.abs(got[i].astype(np.int32) - ref.astype(np.int32))
            assert diff.max() <= 1 and (diff > 0).mean() < 2e-3, (h, w, H, W, diff.max(), (diff > 0).mean())
            assert got[i].max() == 255


def test_postprocess_against_known_answers_of_the_cv2_rule():
    """`uavsal_postprocess` against answers that need neither cv2 nor the restatement (tests/post_vectors.py): linear
    ramps, a one-hot map, both crop branches of utils_data.py:289-303."""
    import post_vectors
    from iip_uavsal_saliency_amd import ops
    for what, pred, R, Cc, exp in post_vectors.cases():
        got = ops.postprocess_predictions(torch.from_numpy(pred)[None, None].cuda(), R, Cc).cpu().numpy()[0].astype(np.int64)
        d = np.abs(got - np.rint(exp).astype(np.int64))
        assert got.shape == (R, Cc) and d.max() <= 1 and (d > 0).mean() < 5e-3, (what, d.max(), (d > 0).mean())
        assert got.max() == 255


def test_predict_video_equals_manual_loop(hip_model, oracle):
    """The streaming driver == the reference caller's loop (Demo_Test.py:65-95) run by hand on the oracle."""
    from iip_uavsal_saliency_amd.stream import predict_video
    from oracle import post_ref
    T, H, W = 2, 72, 104
    u8 = synth.synth_frames_u8(9, H, W)          # 9 frames, time_dims 2 -> 8 kept, groups of 4
    gp = torch.from_numpy(synth.gauss_priors(1, 9, 13))[0]
    op = torch.from_numpy(synth.ob_priors(1, 9, 13))[0]
    hip_model.time_dims, hip_model.precision = T, "f32"
    sal, maps = predict_video(hip_model, torch.from_numpy(u8), gp, op, batch_size=2, return_maps=True)
    assert sal.shape == (8, H, W) and sal.dtype == torch.uint8
    # state kept in the engine's buffer between groups (default) == re-fed state, bit for bit; and a last group
    # of a different length (9 frames of time_dims 2 in groups of 3 chunks: 6 + 2) hands the state to another plan
    sal2, maps2 = predict_video(hip_model, torch.from_numpy(u8), gp, op, batch_size=2, return_maps=True,
                                persistent_state=False)
    assert torch.equal(maps, maps2) and torch.equal(sal, sal2)
    m3a = predict_video(hip_model, torch.from_numpy(u8), gp, op, batch_size=3, return_maps=True)[1]
    m3b = predict_video(hip_model, torch.from_numpy(u8), gp, op, batch_size=3, return_maps=True, persistent_state=False)[1]
    assert torch.equal(m3a, m3b)
    oracle.time_dims = T
    state, ref_maps = None, []
    for i in range(2):
        x = torch.from_numpy(synth.normalize_frames(u8[i * 4:(i + 1) * 4]))
        cb = [gp.unsqueeze(0).repeat(4, 1, 1, 1), op.unsqueeze(0).repeat(4, 1, 1, 1)]
        o, state = oracle(x, cb, state)
        ref_maps.append(o)
    ref_maps = torch.cat(ref_maps, 0)
    assert (maps.cpu() - ref_maps).abs().max().item() <= MAP_TOL["f32"]
    ref0 = post_ref.to_uint8(post_ref.postprocess_predictions(ref_maps[0, 0].numpy(), H, W))
    assert np.abs(sal[0].cpu().numpy().astype(np.int32) - ref0.astype(np.int32)).max() <= 2
    # the result file of the reference's loop (Demo_Test.py:92-95): salmap uint8 [H, W, 1, saved frames], MATLAB v7.3
    import tempfile
    from iip_uavsal_saliency_amd import matio
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "video.mat")
        sal3 = predict_video(hip_model, torch.from_numpy(u8), gp, op, batch_size=2, out_path=path, save_frames=5)
        assert torch.equal(sal3, sal)
        mat = matio.loadmat(path)["salmap"]
        assert mat.dtype == np.uint8 and mat.shape == (H, W, 1, 5)
        assert np.array_equal(mat[:, :, 0, :].transpose(2, 0, 1), sal[:5].cpu().numpy())


@pytest.mark.parametrize("variant", ["twa", "lstm"])
def test_predict_video_overlapped_groups_are_bit_identical(hip_model, variant):
    """`predict_video(overlap=True)`: the groups of one video two deep in flight on two replicas -- everything in front of the
    recurrence of group k + 1 is launched under the tail of group k, only the recurrence waits for (and takes over) the previous
    group's state (`Engine.run_streamed`).  Same maps as the sequential loop of Demo_Test.py:75-86, bit for bit, over 7 groups
    (state carried through both replicas several times); needs the resident state and whole groups."""
    from iip_uavsal_saliency_amd import UAVSAL_LSTM, stream
    m = hip_model
    if variant == "lstm":
        m = UAVSAL_LSTM(time_dims=4)
        synth.load_synth_weights(m, 0)
        m = m.cuda().eval()
    m.time_dims, m.precision = 4, "f32"
    frames = torch.from_numpy(synth.synth_frames_u8(28, 96, 160, 3)).cuda()
    g = torch.from_numpy(synth.gauss_priors(1, 12, 20))[0].cuda()
    o = torch.from_numpy(synth.ob_priors(1, 12, 20, seed=3))[0].cuda()
    if variant == "lstm":      # ((h, c) carried: against a manual loop over forward())
        a_sal, a_maps = stream.predict_video(m, frames, g, o, batch_size=1, return_maps=True, overlap=True)
        b_sal, b_maps = stream.predict_video(m, frames, g, o, batch_size=1, return_maps=True, overlap=True)
        ref, st = [], None
        for k in range(7):
            x = frames[4 * k:4 * k + 4]
            out, s2 = m(x, [g[None].expand(4, -1, -1, -1), o[None].expand(4, -1, -1, -1)], st)
            st = [(s2[0].clone(), s2[1].clone())]
            ref.append(out.clone())
        assert torch.equal(a_maps, torch.cat(ref, 0)) and torch.equal(a_maps, b_maps) and torch.equal(a_sal, b_sal)
        c_sal = stream.predict_video(m, frames, g, o, batch_size=1, overlap=False)      # the one-after-the-other loop carries (h, c) too
        assert torch.equal(c_sal, a_sal)
        return
    seq_sal, seq_maps = stream.predict_video(m, frames, g, o, batch_size=1, return_maps=True, overlap=False)
    ov_sal, ov_maps = stream.predict_video(m, frames, g, o, batch_size=1, return_maps=True, overlap=True)
    auto_sal = stream.predict_video(m, frames, g, o, batch_size=1)            # default: overlapped wherever it applies
    assert torch.equal(auto_sal, seq_sal)
    m.arena_debug = True          # the two ranges of a plan with every released activation NaN-filled behind its last use
    try:
        assert torch.equal(stream.predict_video(m, frames, g, o, batch_size=1, overlap=True), seq_sal)
    finally:
        m.arena_debug = False
    assert torch.equal(seq_maps, ov_maps) and torch.equal(seq_sal, ov_sal) and bool(torch.isfinite(ov_maps).all().item())
    ov2 = stream.predict_video(m, frames[:24], g, o, batch_size=2, overlap=True)          # groups of 2 x time_dims frames
    seq2 = stream.predict_video(m, frames[:24], g, o, batch_size=2, overlap=False)
    assert torch.equal(stream.predict_video(m, frames[:28], g, o, batch_size=2), stream.predict_video(m, frames[:28], g, o, batch_size=2, overlap=False))
    assert torch.equal(ov2, seq2)
    # the second handle is kept on the model between videos and must follow an in-place weight edit (both handles rebuild their plans
    # from the new values: same maps as the one-after-the-other loop, different from before)
    with torch.no_grad():
        m.conv_out_st.conv[3].bias.add_(0.5)
    try:
        ov3 = stream.predict_video(m, frames, g, o, batch_size=1, overlap=True)
        seq3 = stream.predict_video(m, frames, g, o, batch_size=1, overlap=False)
        assert torch.equal(ov3, seq3) and bool((ov3 != ov_sal).any().item())
    finally:
        with torch.no_grad():
            m.conv_out_st.conv[3].bias.sub_(0.5)
    # a shorter last group follows the overlapped whole groups on a plan of its own, with their state
    assert torch.equal(stream.predict_video(m, frames[:28], g, o, batch_size=2, overlap=True),
                       stream.predict_video(m, frames[:28], g, o, batch_size=2, overlap=False))
    with pytest.raises(RuntimeError):          # fewer than two whole groups: nothing to overlap
        stream.predict_video(m, frames[:12], g, o, batch_size=2, overlap=True)
    with pytest.raises(RuntimeError):
        stream.predict_video(m, frames, g, o, batch_size=1, overlap=True, persistent_state=False)


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
def test_lstm_variant_vs_reference_golden_and_oracle(golden_dir, prec):
    """UAVSAL_LSTM (reference model.py:960-1076): ConvLSTM recurrence, (h, c) carried across two calls."""
    from iip_uavsal_saliency_amd import UAVSAL_LSTM
    g = np.load(os.path.join(golden_dir, "e2e_lstm_96x160_T4_two_calls.npz"))
    m = UAVSAL_LSTM(time_dims=4, precision=prec)
    synth.load_synth_weights(m, int(g["seed"]))
    m = m.cuda().eval()
    state = None
    for c in range(2):
        x, cb = make_inputs(4, 96, 160, int(g["seed"]), t0=c * 4)
        out, st = m(x.cuda(), [cb[0].cuda(), cb[1].cuda()], state)
        state = [(st[0], st[1])]
        sfx = "" if c == 0 else f"_call{c}"
        assert np.abs(out.cpu().numpy() - g["out" + sfx]).max() <= MAP_TOL[prec]
        ss = int(g["state_stride"])
        assert np.abs(st[0].cpu().contiguous().view(-1).numpy()[::ss] - g["state" + sfx]).max() <= STATE_TOL[prec]
        assert np.abs(st[1].cpu().contiguous().view(-1).numpy()[::ss] - g["cstate" + sfx]).max() <= STATE_TOL[prec]


def test_inputs_are_read_in_place_and_never_modified(hip_model):
    """The launch loop binds the caller's tensors (no staging copy): contiguous and non-contiguous inputs give the
    same result, and nothing the caller passed is written to (SURVEY.md 8(b) Ownership)."""
    x, cb = make_inputs(4, 96, 160)
    hip_model.time_dims, hip_model.precision = 4, "f32"
    xd, g, o = x.cuda(), cb[0].cuda(), cb[1].cuda()
    st = torch.rand((1, 256, 12, 20), generator=torch.Generator().manual_seed(9)).cuda()
    keep = [t.clone() for t in (xd, g, o, st)]
    a, sa = hip_model(xd, [g, o], [st])
    # non-contiguous views of the same values: channels-last frames, a strided slice of a wider prior tensor
    x_cl = xd.to(memory_format=torch.channels_last)
    g_wide = torch.zeros((4, 16, 12, 20), device="cuda")
    g_wide[:, ::2] = g
    b, sb = hip_model(x_cl, [g_wide[:, ::2], o], [st.permute(0, 1, 3, 2).contiguous().permute(0, 1, 3, 2)])
    assert torch.equal(a, b) and torch.equal(sa[0], sb[0])
    for t, k in zip((xd, g, o, st), keep):
        assert torch.equal(t, k)


def test_in_place_weight_edit_rebuilds_the_plan(hip_model):
    """Engines are built from the parameter values; a version-bumping in-place edit (an op under no_grad, an optimizer
    step, load_state_dict) is detected by the version scan and the plan is rebuilt instead of silently using stale
    packed weights.  Edits through `param.data` do not bump the version: those need model.invalidate_engines()."""
    x, cb = make_inputs(4, 96, 160)
    hip_model.time_dims, hip_model.precision = 4, "f32"
    args = (x.cuda(), [cb[0].cuda(), cb[1].cuda()], None)
    a, _ = hip_model(*args)
    wt = hip_model.conv_out_st.conv[3].weight          # BatchNorm gamma of the decoder's projection
    old = wt.detach().clone()
    try:
        with torch.no_grad():
            wt.mul_(0.5)
        b, _ = hip_model(*args)
        assert (a - b).abs().max().item() > 1e-3         # the new weights were used
    finally:
        with torch.no_grad():
            wt.copy_(old)
    c, _ = hip_model(*args)
    assert torch.equal(a, c)


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["f32", "f16x3"])
def test_request_pipeline_equals_sequential_calls(prec):
    """Independent requests two deep in flight on two replicas / host streams == the same requests one after another."""
    from iip_uavsal_saliency_amd import UAVSal, synth
    from iip_uavsal_saliency_amd.stream import RequestPipeline
    dev = torch.device("cuda:0")
    m = UAVSal(time_dims=4, precision=prec)
    synth.load_synth_weights(m, 0)
    m = m.to(dev).eval()
    reqs = []
    for k in range(5):
        g = torch.Generator().manual_seed(300 + k)
        x = torch.rand((1, 4, 3, 96, 160), generator=g).to(dev)
        cb = [torch.rand((1, 4, 8, 12, 20), generator=g).to(dev), torch.rand((1, 4, 20, 12, 20), generator=g).to(dev)]
        st = torch.rand((1, 256, 12, 20), generator=g).to(dev)
        reqs.append((x, cb, st))
    want = [m.forward_clips(x, cb, st) for x, cb, st in reqs]
    torch.cuda.synchronize(dev)
    w0 = want[0][0].clone()
    pipe = RequestPipeline(m, streams=2)
    assert pipe.models[1]._wshared is m._wshared and pipe.models[1]._engines is not m._engines
    got = [pipe.forward_clips(x, cb, st) for x, cb, st in reqs]
    pipe.synchronize()
    for (wo, ws), (go, gs, _) in zip(want, got):
        assert torch.equal(wo, go) and torch.equal(ws, gs)
    # the in-flight handles run without lanes (stream._inflight_replicas) and keep their plans from round to round
    assert all(not r.use_lanes for r in pipe.models) and m.use_lanes
    plans = [next(iter(r._engines.values())) for r in pipe.models]
    got = [pipe.forward_clips(x, cb, st) for x, cb, st in reqs[:2]]
    pipe.synchronize()
    assert [next(iter(r._engines.values())) for r in pipe.models] == plans
    # an in-place weight edit reaches the handles although the model itself does not run in between
    with torch.no_grad():
        next(m.parameters()).mul_(0.5)
    got = [pipe.forward_clips(x, cb, st) for x, cb, st in reqs[:3]]
    pipe.synchronize()
    assert [next(iter(r._engines.values())) for r in pipe.models] != plans
    want = [m.forward_clips(x, cb, st) for x, cb, st in reqs[:3]]
    torch.cuda.synchronize(dev)
    for (wo, ws), (go, gs, _) in zip(want, got):
        assert torch.equal(wo, go) and torch.equal(ws, gs)
    assert not torch.equal(want[0][0], w0)


@pytest.mark.gpu
def test_deepcopy_and_whole_model_save_after_a_forward():
    """A model that has run (plans recorded, weights packed, stream handles cached) can be deep-copied and saved whole, as the
    reference saves its checkpoints (model.py:339); the copy builds its own plans and gives the same maps."""
    import copy
    import io
    from iip_uavsal_saliency_amd import UAVSal, synth
    from iip_uavsal_saliency_amd.stream import predict_video
    dev = torch.device("cuda:0")
    m = UAVSal(time_dims=4)
    synth.load_synth_weights(m, 0)
    m = m.to(dev).eval()
    g = torch.Generator().manual_seed(11)
    x = torch.rand((8, 3, 96, 160), generator=g).to(dev)
    cb = [torch.rand((8, 8, 12, 20), generator=g).to(dev), torch.rand((8, 20, 12, 20), generator=g).to(dev)]
    want, wst = m(x, cb, None)
    u8 = (torch.rand((16, 3, 96, 160), generator=g) * 255).to(torch.uint8).to(dev)
    sal = predict_video(m, u8, cb[0][0], cb[1][0], batch_size=1)          # overlapped: replicas and streams cached on the model
    assert "_stream_replicas" in m.__dict__
    buf = io.BytesIO()
    torch.save(m, buf)
    buf.seek(0)
    for d in (copy.deepcopy(m), torch.load(buf, weights_only=False)):
        assert len(d._engines) == 0 and "_stream_replicas" not in d.__dict__
        got, gst = d(x, cb, None)
        assert torch.equal(got, want) and torch.equal(gst[0], wst[0])
        assert torch.equal(predict_video(d, u8, cb[0][0], cb[1][0], batch_size=1), sal)
    got, _ = m(x, cb, None)                   # the original keeps working
    assert torch.equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("overlap", [False, True])
def test_predict_video_from_host_frames_equals_device_frames(overlap):
    """Frames in host memory (pinned or pageable) are uploaded group by group on a copy stream ahead of the launches: same maps,
    bit for bit, as with the whole video resident on the device."""
    from iip_uavsal_saliency_amd import UAVSal, synth
    from iip_uavsal_saliency_amd.stream import predict_video
    dev = torch.device("cuda:0")
    m = UAVSal(time_dims=4)
    synth.load_synth_weights(m, 0)
    m = m.to(dev).eval()
    g = torch.Generator().manual_seed(77)
    u8 = (torch.rand((26, 3, 96, 160), generator=g) * 255).to(torch.uint8)          # 6 groups of 4 + 2 dropped frames
    gp, op_ = torch.rand((8, 12, 20), generator=g), torch.rand((20, 12, 20), generator=g)
    want, wmaps = predict_video(m, u8.to(dev), gp, op_, batch_size=1, overlap=overlap, return_maps=True)
    assert want.shape[0] == 24
    for host in (u8, u8.pin_memory()):
        got, gmaps = predict_video(m, host, gp, op_, batch_size=1, overlap=overlap, return_maps=True)
        assert torch.equal(got, want) and torch.equal(gmaps, wmaps)
    if not overlap:                           # a shorter last group (batch_size 4 -> 16 + 8 frames) from host memory
        want2 = predict_video(m, u8.to(dev), gp, op_, batch_size=4)
        assert torch.equal(predict_video(m, u8.pin_memory(), gp, op_, batch_size=4), want2)


@pytest.mark.gpu
@pytest.mark.parametrize("cls_name", ["UAVSal", "UAVSAL_LSTM"])
def test_predict_video_overlaps_whole_groups_and_finishes_a_ragged_tail(cls_name):
    """46 frames, time_dims 4, batch_size 2: 11 chunks = 5 whole groups of 8 frames (overlapped, two in flight) + one group of 4
    frames that runs afterwards on its own plan with the state of the fifth -- the maps of the reference's strictly sequential loop."""
    import iip_uavsal_saliency_amd as pkg
    from iip_uavsal_saliency_amd import synth
    from iip_uavsal_saliency_amd.stream import predict_video
    dev = torch.device("cuda:0")
    m = getattr(pkg, cls_name)(time_dims=4)
    synth.load_synth_weights(m, 0)
    m = m.to(dev).eval()
    g = torch.Generator().manual_seed(5)
    u8 = (torch.rand((46, 3, 96, 160), generator=g) * 255).to(torch.uint8)
    gp, op_ = torch.rand((8, 12, 20), generator=g), torch.rand((20, 12, 20), generator=g)
    want, wmaps = predict_video(m, u8.to(dev), gp, op_, batch_size=2, overlap=False, return_maps=True)
    assert want.shape[0] == 44
    for src in (u8.to(dev), u8.pin_memory()):
        got, gmaps = predict_video(m, src, gp, op_, batch_size=2, return_maps=True)          # overlap=None: applies (5 whole groups)
        assert "_stream_replicas" in m.__dict__
        assert torch.equal(gmaps, wmaps) and torch.equal(got, want)
    with pytest.raises(RuntimeError, match="at least two whole groups"):
        predict_video(m, u8[:12].to(dev), gp, op_, batch_size=2, overlap=True)              # one whole group + a tail

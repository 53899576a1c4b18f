"""The oracle (CPU restatement, oracle/uavsal_ref.py) against the golden vectors that
oracle/make_goldens.py produced by running the reference's own model.py on CPU."""
import os

import numpy as np
import pytest
import torch

from iip_uavsal_saliency_amd import synth
from oracle import uavsal_ref as R

CASES = ["e2e_96x160_T4", "e2e_96x160_B4T5", "e2e_96x160_T4_two_calls", "e2e_72x104_T3"]
# constructor values other than the Demo default: priors dropped one by one, and none (reference model.py:281-324, 346-365)
BIAS = ["e2e_96x160_T4_bias101", "e2e_96x160_B2T4_bias010", "e2e_96x160_T4_bias000_two_calls", "e2e_96x160_B2T4_bias001"]
BIG = ["e2e_288x512_T8",
       # Demo_Test.py's own call at its real size: one forward of 4 x 5 = 20 frames at 360x640, twice (carried state)
       "e2e_360x640_B4T5_two_calls"]


def make_inputs(n, H, W, seed=0, t0=0):
    h, w = H // 8, W // 8
    x = torch.from_numpy(synth.normalize_frames(synth.synth_frames_u8(n, H, W, seed, t0)))
    cb = [torch.from_numpy(synth.gauss_priors(n, h, w)), torch.from_numpy(synth.ob_priors(n, h, w, seed=seed))]
    return x, cb


def _sub(t, stride):
    return t.contiguous().view(-1).numpy()[::stride]


def test_oracle_lstm_matches_reference_golden(golden_dir):
    """The ConvLSTM variant (reference UAVSAL_LSTM, model.py:960-1076) incl. the (h, c) state protocol."""
    g = np.load(os.path.join(golden_dir, "e2e_lstm_96x160_T4_two_calls.npz"))
    model = R.build_oracle(time_dims=4, seed=int(g["seed"]), rnn="lstm")
    state = None
    for c in range(2):
        x, cb = make_inputs(4, 96, 160, int(g["seed"]), t0=c * 4)
        out, st = model(x, cb, state)
        state = [(st[0], st[1])]
        sfx = "" if c == 0 else f"_call{c}"
        np.testing.assert_allclose(out.numpy(), g["out" + sfx], rtol=0, atol=2e-6)
        np.testing.assert_allclose(_sub(st[0], int(g["state_stride"])), g["state" + sfx], rtol=0, atol=2e-5)
        np.testing.assert_allclose(_sub(st[1], int(g["state_stride"])), g["cstate" + sfx], rtol=0, atol=2e-5)


@pytest.mark.parametrize("name", CASES + BIAS + BIG)
def test_oracle_matches_reference_golden(name, golden_dir):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    H, W, T, B = int(g["H"]), int(g["W"]), int(g["T"]), int(g["B"])
    bias = tuple(int(b) for b in g["bias_type"]) if "bias_type" in g else (1, 1, 1)
    model = R.build_oracle(time_dims=T, seed=int(g["seed"]), bias_type=bias)
    n = B * T
    state = None
    for c in range(int(g["calls"])):
        x, cb = make_inputs(n, H, W, int(g["seed"]), t0=c * n)
        taps = {}
        out, st = model(x, cb, state, taps)
        state = [st[0]]
        sfx = "" if c == 0 else f"_call{c}"
        # same ATen CPU operators as the reference -> agreement to fp32 round-off
        np.testing.assert_allclose(out.numpy(), g["out" + sfx], rtol=0, atol=2e-6)
        np.testing.assert_allclose(taps["logits"].numpy(), g["logits" + sfx], rtol=0, atol=5e-5)
        np.testing.assert_allclose(_sub(st[0], int(g["state_stride"])), g["state" + sfx], rtol=0, atol=2e-5)
        if c == 0:
            for k in ("sfnet", "st0", "st1", "fust_in_cb", "prefuse", "rnn"):
                if not any(bias) and k in ("fust_in_cb", "prefuse"):
                    assert "tap_" + k not in g          # no fusion blocks in a model without priors
                    continue
                np.testing.assert_allclose(_sub(taps[k], int(g["tap_stride"])), g["tap_" + k], rtol=0, atol=5e-5)


def test_oracle_clips_match_reference_golden(golden_dir):
    """Batched-clips goldens (BASELINE configs[2]/[3] shape: 8 independent reference calls at 360x640): the
    oracle's forward_clips on one of the clips (CPU time) against the reference's maps and state digest."""
    g = np.load(os.path.join(golden_dir, "clips_360x640_C8_T8.npz"))
    H, W, T, C, seed = int(g["H"]), int(g["W"]), int(g["T"]), int(g["C"]), int(g["seed"])
    assert int(g["map_stride"]) == 1
    c = 5
    h, w = H // 8, W // 8
    x = torch.from_numpy(synth.normalize_frames(synth.synth_frames_u8(T, H, W, seed + c)))[None]
    cb = [torch.from_numpy(synth.gauss_priors(T, h, w))[None], torch.from_numpy(synth.ob_priors(T, h, w, seed=seed + c))[None]]
    out, st = R.build_oracle(time_dims=T, seed=seed).forward_clips(x, cb)
    ref = g["out"].reshape(C, T, 1, h, w)[c]
    np.testing.assert_allclose(out[0].numpy(), ref, rtol=0, atol=2e-6)
    assert np.array_equal(g["out"].reshape(C, T, 1, h, w)[0], g["out_clip0"])


def test_state_dict_schema_known_answer():
    """Reference known answer: 51.59 MB params+buffers (Tools/Getmodelsize_demo.py:93);
    SURVEY.md 8(b): 685 entries, 13 407 338 parameters."""
    m = R.RefUAVSal()
    sd = m.state_dict()
    assert len(sd) == 685
    assert sum(p.numel() for p in m.parameters()) == 13407338
    mb = (sum(p.numel() * p.element_size() for p in m.parameters())
          + sum(b.numel() * b.element_size() for b in m.buffers())) / 1024 / 1024
    assert "%.2f" % mb == "51.59"
    assert sd["rnn.cell_list.0.rnn_conv.weight"].shape == (256, 512, 3, 3)
    assert sd["sfnet.features.features.18.0.weight"].shape == (1280, 320, 1, 1)


def test_context_tiling_quirk():
    """model.py:361: frame k gets the context of chunk k mod B (not k // T)."""
    g_dir = os.path.join(os.path.dirname(__file__), "golden")
    model = R.build_oracle(time_dims=4)
    x, cb = make_inputs(8, 96, 160)
    taps = {}
    model(x, cb, None, taps)
    ctx_in = taps["fust_in_cb"]          # depends on ctx through fucb_layer
    assert ctx_in.shape[0] == 8


def test_temporal_difference_edges():
    x1 = torch.arange(5, dtype=torch.float32).view(5, 1, 1, 1) ** 2
    d = R.temporal_differences(x1).view(5, 2)
    # frame 0: [x1-x0, x0-x1]; middle: [xi-x(i-1), xi-x(i+1)]; last: [xl-x(l-1), x(l-1)-xl]
    assert d[0].tolist() == [1.0, -1.0]
    assert d[2].tolist() == [3.0, -5.0]
    assert d[4].tolist() == [7.0, -7.0]
    with pytest.raises(RuntimeError):
        R.temporal_differences(x1[:1])


def test_convlstm_step_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "convlstm_step.npz"))
    hid, (h, w), seed = int(g["hid"]), g["hw"], int(g["seed"])
    wgt = torch.from_numpy(synth.synth_tensor("convlstm.rnn_conv.weight", (4 * hid, 2 * hid, 3, 3), seed))
    mk = lambda nm: torch.from_numpy(synth.hash_normal(nm, hid * h * w, seed).astype(np.float32)).view(1, hid, h, w)
    hn, cn = R.convlstm_cell_step(wgt, mk("convlstm.x"), mk("convlstm.h"), mk("convlstm.c"))
    np.testing.assert_allclose(hn.numpy(), g["h_next"], atol=1e-6)
    np.testing.assert_allclose(cn.numpy(), g["c_next"], atol=1e-6)


def test_forward_clips_equals_independent_calls():
    model = R.build_oracle(time_dims=3)
    x, cb = make_inputs(6, 72, 104)
    xc = x.view(2, 3, 3, 72, 104)
    cbc = [cb[0].view(2, 3, 8, 9, 13), cb[1].view(2, 3, 20, 9, 13)]
    out, st = model.forward_clips(xc, cbc)
    o0, s0 = model(x[:3], [cb[0][:3], cb[1][:3]], None)
    o1, s1 = model(x[3:], [cb[0][3:], cb[1][3:]], None)
    assert torch.equal(out[0], o0) and torch.equal(out[1], o1)
    assert torch.equal(st[0:1], s0[0]) and torch.equal(st[1:2], s1[0])


def test_post_ref_against_known_answers_of_the_cv2_rule():
    """the post-processing restatement (oracle/post_ref.py) against answers that follow from cv2's
    documented INTER_LINEAR mapping and the crop arithmetic of utils_data.py:289-303 (tests/post_vectors.py) --
    linear ramps (exact under linear interpolation, incl. the clamped borders), a one-hot map (the four weights
    0.25 / 0.75 / 0.75 / 0.25), one rows_rate > cols_rate and one opposite crop."""
    import post_vectors
    from oracle import post_ref
    for what, pred, R, Cc, exp in post_vectors.cases():
        got = post_ref.postprocess_predictions(pred, R, Cc)
        assert got.shape == (R, Cc), what
        assert np.abs(got.astype(np.float64) - exp).max() < 2e-3, (what, np.abs(got - exp).max())      # of 255
        q = post_ref.to_uint8(got).astype(np.int64)
        d = np.abs(q - np.rint(exp).astype(np.int64))
        assert d.max() <= 1 and (d > 0).mean() < 5e-3, (what, d.max(), (d > 0).mean())       # .5 boundaries only
        assert q.max() == 255
    # the host-side resize used for the priors is the same rule
    from iip_uavsal_saliency_amd import priors
    pred = post_vectors.ramp_case(45, 80, 360, 640)[0]
    assert np.array_equal(priors.resize_linear(pred, 360, 640), post_ref._resize_linear(pred, 360, 640))


def test_mat_writer_round_trip_and_layout(tmp_path):
    """The result file of the caller loop (Demo_Test.py:93-95): uint8 `salmap` [H,W,1,F] as MATLAB v7.3.  Round trip through
    the reader, and the container laid out like the reference's own .mat files (512-byte MATLAB header, superblock v0 at
    512 with base address 512, MATLAB_class attribute, reversed dimension order)."""
    from iip_uavsal_saliency_amd import matio
    rng = np.random.default_rng(5)
    sal = rng.integers(0, 256, (36, 64, 1, 7), dtype=np.uint8)
    aux = rng.standard_normal((5, 3)).astype(np.float32)
    path = str(tmp_path / "out.mat")
    matio.savemat(path, {"salmap": sal, "aux": aux, "d": np.arange(6.0).reshape(2, 3)})
    got = matio.loadmat(path)
    assert sorted(got) == ["aux", "d", "salmap"]
    assert got["salmap"].dtype == np.uint8 and np.array_equal(got["salmap"], sal)
    assert got["aux"].dtype == np.float32 and np.array_equal(got["aux"], aux) and np.array_equal(got["d"], np.arange(6.0).reshape(2, 3))
    raw = open(path, "rb").read()
    assert raw[:20] == b"MATLAB 7.3 MAT-file," and raw[124:128] == b"\x00\x02IM" and raw[512:520] == b"\x89HDF\r\n\x1a\n"
    import struct
    base, _, eof, _ = struct.unpack_from("<QQQQ", raw, 512 + 24)
    assert base == 512 and eof == len(raw)
    assert b"MATLAB_class\x00" in raw and b"uint8" in raw and b"single" in raw and b"double" in raw
    # HDF5 stores the dimensions reversed: the dataspace of salmap is (7, 1, 64, 36) and the bytes are column-major
    assert struct.pack("<QQQQ", 7, 1, 64, 36) in raw and sal.tobytes(order="F") in raw
    ref = "/root/reference/gauss_priors.mat"
    if os.path.exists(ref):      # same superblock / root-group prefix as a file hdf5storage wrote (up to the end-of-file address)
        r = open(ref, "rb").read()
        assert r[512:512 + 40] == raw[512:512 + 40] and r[512 + 48:512 + 96 + 40] == raw[512 + 48:512 + 96 + 40]
    if os.path.exists(ref):
        # ... and message by message against the dataset header hdf5storage wrote for the reference's own priors file: write the SAME
        # array under the same name and compare the object-header messages that do not depend on the storage layout (the reference
        # file is chunked + shuffle + deflate, ours contiguous): datatype (0x03) and the MATLAB_class attribute (0x0c) byte for
        # byte, the dataspace's (0x01) rank and dimensions, the fill-value message's (0x05) version / defined / size fields
        def dataset_msgs(p_, name):
            raw_ = open(p_, "rb").read()
            h5 = matio._H5(raw_)
            ent = h5.group_entries(h5.root_btree, h5.root_heap)
            return [(t, raw_[pos:pos + sz]) for (t, pos, sz) in h5.messages(ent[name])]
        maps = matio.loadmat(ref)["PriorMaps"]
        p2 = str(tmp_path / "priors_again.mat")
        matio.savemat(p2, {"PriorMaps": maps})
        theirs, ours = dataset_msgs(ref, "PriorMaps"), dataset_msgs(p2, "PriorMaps")
        first = lambda ms, t: [m for (tt, m) in ms if tt == t][0]
        assert first(ours, 0x03) == first(theirs, 0x03)
        cls_theirs = [m for (tt, m) in theirs if tt == 0x0c and b"MATLAB_class" in m][0]
        assert first(ours, 0x0c) == cls_theirs
        ds_o, ds_t = first(ours, 0x01), first(theirs, 0x01)
        assert ds_o[0] == ds_t[0] == 1 and ds_o[1] == ds_t[1] == maps.ndim and ds_o[8:8 + 8 * maps.ndim] == ds_t[8:8 + 8 * maps.ndim]
        fv_o, fv_t = first(ours, 0x05), first(theirs, 0x05)
        assert fv_o[0] == fv_t[0] == 2 and fv_o[3] == fv_t[3] == 1 and fv_o[4:8] == fv_t[4:8]
        assert np.array_equal(matio.loadmat(p2)["PriorMaps"], maps)
    with pytest.raises(ValueError):
        matio.savemat(path, {"x": np.zeros(3, dtype=np.complex64)})
    # an INDEPENDENT reader, where one is installed (none in the build container: the check above against the reference's own
    # file is then all there is): the consumers of the reference's result files are hdf5storage / MATLAB
    try:
        import h5py
    except ImportError:
        h5py = None
    if h5py is not None:
        with h5py.File(path, "r") as f:
            assert sorted(f.keys()) == ["aux", "d", "salmap"]
            assert f["salmap"].shape == (7, 1, 64, 36) and f["salmap"].dtype == np.uint8
            assert np.array_equal(np.asarray(f["salmap"]).transpose(3, 2, 1, 0), sal)
            assert f["salmap"].attrs["MATLAB_class"] == b"uint8" and f["aux"].attrs["MATLAB_class"] == b"single"
            assert np.array_equal(np.asarray(f["aux"]).T, aux)

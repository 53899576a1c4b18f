"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol the header
declares, the host packing is what the kernels document, the drop-in module tree has the
reference's schema, errors surface as in the reference, and the clip sharding / all-gather
layer is correct for world_size 2 over gloo."""
import json
import os
import re
import socket
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from iip_uavsal_saliency_amd import packing as P
from iip_uavsal_saliency_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from iip_uavsal_saliency_amd import build, _lib
    build.build()                      # hipcc cross-compiles gfx950 without a GPU
    return _lib.load()


def test_library_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "uavsal_hip.h")).read()
    declared = set(re.findall(r"\b(uavsal_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"uavsal_plan"}        # the opaque type
    assert len(declared) >= 24
    from iip_uavsal_saliency_amd import _lib
    bound = {s[0] for s in _lib.SYMBOLS}
    assert declared == bound, (declared ^ bound)
    for name in declared:
        assert hasattr(lib, name)
    assert lib.uavsal_abi_version() == 20
    assert b"gfx950" in lib.uavsal_build_info()


def test_argument_validation_without_gpu(lib):
    """Rejected descriptors return before any HIP call, so this runs without a device."""
    import ctypes as C
    from iip_uavsal_saliency_amd import _lib as L
    d = L.ConvDesc()
    assert lib.uavsal_conv_gemm(C.byref(d), None) == -1            # null pointers
    d.a, d.w, d.out = 16, 16, 16
    d.n_img, d.H, d.W, d.Cin, d.Cout, d.taps, d.lda, d.ldc = 1, 4, 4, 6, 8, 1, 6, 8
    d.a_img_stride = d.o_img_stride = 16
    assert lib.uavsal_conv_gemm(C.byref(d), None) == -2            # Cin % 4
    d.Cin, d.lda, d.taps = 16, 16, 9
    assert lib.uavsal_conv_gemm(C.byref(d), None) == -3            # 3x3 needs Cin % 32
    t = L.TdiffDesc()
    t.inp, t.out, t.n_img, t.HW, t.C, t.seq_len, t.ldi, t.ldo = 16, 16, 1, 4, 32, 1, 32, 64
    assert lib.uavsal_tdiff(C.byref(t), None) == -3                # one frame per sequence
    p = lib.uavsal_plan_create()
    assert lib.uavsal_plan_size(p) == 0
    assert lib.uavsal_plan_graph_launch(p, None) == -4             # no graph built yet
    assert lib.uavsal_plan_status(p, 1) == 0                       # nothing was run
    g = L.GuardDesc()
    assert lib.uavsal_guard(C.byref(g), None) == -1                # no error word
    c = L.CopyDesc()
    c.inp, c.out, c.rows, c.row_floats, c.in_pitch, c.out_pitch = 16, 16, 2, 6, 8, 8
    assert lib.uavsal_copy_rows(C.byref(c), None) == -2            # rows must be float4 multiples
    lib.uavsal_plan_destroy(p)


def test_fold_bn_matches_batch_norm():
    bn = torch.nn.BatchNorm2d(24).eval()
    g = torch.Generator().manual_seed(0)
    bn.weight.data = torch.rand(24, generator=g) + 0.5
    bn.bias.data = torch.rand(24, generator=g) - 0.5
    bn.running_mean.data = torch.rand(24, generator=g) - 0.5
    bn.running_var.data = torch.rand(24, generator=g) + 0.1
    x = torch.rand((2, 24, 5, 5), generator=g) * 4 - 2
    s, b = P.fold_bn(bn)
    assert torch.allclose(x * s.view(1, -1, 1, 1) + b.view(1, -1, 1, 1), bn(x), atol=1e-6)


@pytest.mark.parametrize("shape", [(24, 20, 1, 1), (1, 1536, 1, 1), (40, 64, 3, 3)])
def test_pack_conv_weight_layouts(shape):
    g = torch.Generator().manual_seed(1)
    w = torch.rand(shape, generator=g) - 0.5
    cout, cin, kh, _ = shape
    k = cin * kh * kh

    def k_order(kt):     # 1x1: k = ci; 3x3: k = ((ci // kt) * 9 + tap) * kt + ci % kt  (uavsal_hip.h)
        if kh == 1:
            return w.reshape(cout, k)
        return w.reshape(cout, cin // kt, kt, 9).permute(0, 1, 3, 2).reshape(cout, k)

    ref = k_order(16)
    f = P.pack_conv_weight(w, "f32").view(torch.float32)
    npad, kpad = P.roundup(cout, 32), P.roundup(k, 16)
    f = f.view(npad, kpad)
    assert torch.equal(f[:cout, :k], ref) and f[cout:].abs().sum() == 0 and f[:, k:].abs().sum() == 0
    ref = k_order(32)
    kpad = P.roundup(k, 32)
    inv = torch.empty(32, dtype=torch.long)
    inv[torch.tensor(P._K_PERM32)] = torch.arange(32)
    for prec, dt, scale, tol in (("bf16x3", torch.bfloat16, 1.0, 2.0 ** -15), ("f16x3", torch.float16, 64.0, 2.0 ** -20)):
        hl = P.pack_conv_weight(w, prec).view(dt).view(kpad // 32, 2, npad, 32).float()   # [step][hi|lo][n][32]
        rec = (hl[:, 0] + hl[:, 1]).permute(1, 0, 2)[:, :, inv].reshape(npad, kpad) / scale
        assert (rec[:cout, :k] - ref).abs().max().item() <= tol * ref.abs().max().item()
    h1 = P.pack_conv_weight(w, "bf16").view(torch.bfloat16).view(kpad // 32, npad, 32)
    assert torch.equal(h1, P.pack_conv_weight(w, "bf16x3").view(torch.bfloat16).view(kpad // 32, 2, npad, 32)[:, 0])


def test_pack_dw_and_stem():
    w = torch.arange(4 * 9, dtype=torch.float32).view(4, 1, 3, 3)
    p = P.pack_dw_weight(w)
    assert p.shape == (9, 4) and p[5, 2] == w[2, 0, 1, 2]
    ws = torch.arange(32 * 27, dtype=torch.float32).view(32, 3, 3, 3)
    s = P.pack_stem_weight(ws)
    assert s.shape == (27, 32) and s[1 * 9 + 2 * 3 + 1, 7] == ws[7, 1, 2, 1]


@pytest.mark.parametrize("r", [2, 4])
def test_pack_wino_weight_is_the_winograd_filter_transform(r):
    """packing.pack_wino_weight: the (r + 2)^2 filter matrices (G g G^T)_k in the layout of the grouped GEMM
    (uavsal_conv_desc.w_group_stride).  Checked through the Winograd identity on the CPU in fp64:
    A^T [ sum_c U_k .* (B^T d B)_k ] A == conv3x3(d, g) for one r x r output tile (csrc/winograd.hip uses the same
    B^T / A^T; the tile transforms are restated here from Lavin & Gray, points 0, +-1 (r = 2) and 0, +-1, +-2 (r = 4))."""
    bt = {2: [[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]],
          4: [[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
              [0, 4, 0, -5, 0, 1]]}[r]
    at = {2: [[1, 1, 1, 0], [0, 1, -1, -1]],
          4: [[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]]}[r]
    bt, at = torch.tensor(bt, dtype=torch.float64), torch.tensor(at, dtype=torch.float64)
    pp = r + 2
    g = torch.Generator().manual_seed(7)
    cin, cout = 40, 24                                   # not multiples of 32: the packing pads both
    w = torch.randn((cout, cin, 3, 3), generator=g, dtype=torch.float64)
    d = torch.randn((cin, pp, pp), generator=g, dtype=torch.float64)
    packed = P.pack_wino_weight(w.float(), r)
    kpad, npad = P.roundup(cin, 32), P.roundup(cout, 32)
    u = packed.view(torch.float32).reshape(pp * pp, npad, kpad).double()
    assert torch.count_nonzero(u[:, cout:, :]) == 0 and torch.count_nonzero(u[:, :, cin:]) == 0
    v = torch.einsum("ij,cjk,lk->cil", bt, d, bt).reshape(cin, pp * pp)                  # (B^T d B)_k per channel
    m = torch.einsum("koc,ck->ok", u[:, :cout, :cin], v).reshape(cout, pp, pp)          # the GEMM, per plane k
    y = torch.einsum("ij,ojk,lk->oil", at, m, at)                                        # A^T m A
    ref = F.conv2d(d[None], w)[0]                                                        # valid conv of the patch = r x r outputs
    assert tuple(y.shape) == tuple(ref.shape) == (cout, r, r)
    assert (y - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()                 # U was rounded to fp32 once


def test_dropin_schema_matches_reference():
    """Same keys/shapes as the oracle tree (itself pinned to the reference's 51.59 MB known
    answer, Tools/Getmodelsize_demo.py:93) and the attribute names that tool touches (:52-82)."""
    from iip_uavsal_saliency_amd import UAVSal
    from oracle.uavsal_ref import RefUAVSal
    m, r = UAVSal(), RefUAVSal()
    sd, rd = m.state_dict(), r.state_dict()
    assert list(sd.keys()) == list(rd.keys()) and len(sd) == 685
    assert all(sd[k].shape == rd[k].shape for k in sd)
    assert sum(p.numel() for p in m.parameters()) == 13407338
    for attr in ("sfnet", "st_layer", "fust_layer", "gauss_cb_layer", "ob_cb_layer", "cxt_cb_prior",
                 "fucb_layer", "fucbst_layer", "rnn", "conv_out_st", "time_dims", "num_stblock", "num_cb"):
        assert hasattr(m, attr)
    assert m.rnn.cell_list[0].rnn_conv.weight.shape == (256, 512, 3, 3)
    m.load_state_dict(r.state_dict())          # weights interchange with the reference schema


def test_no_cpu_fallback():
    from iip_uavsal_saliency_amd import UAVSal
    m = UAVSal(time_dims=2).eval()
    x = torch.zeros(2, 3, 72, 104)
    cb = [torch.zeros(2, 8, 9, 13), torch.zeros(2, 20, 9, 13)]
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(x, cb, None)
    with pytest.raises(RuntimeError):
        m.sfnet(x)                              # sub-modules hold parameters only
    m.train()
    with pytest.raises(RuntimeError, match="inference-only"):
        m(x, cb, None)


def test_synth_is_deterministic():
    a = synth.synth_tensor("sfnet.conv_last.0.weight", (256, 448, 3, 3))
    b = synth.synth_tensor("sfnet.conv_last.0.weight", (256, 448, 3, 3))
    assert np.array_equal(a, b) and abs(float(a.std()) - np.sqrt(2.0 / (448 * 9))) < 1e-3
    f = synth.synth_frames_u8(3, 24, 40)
    assert f.dtype == np.uint8 and f.shape == (3, 3, 24, 40) and not np.array_equal(f[0], f[1])
    g = synth.gauss_priors(2, 45, 80)
    assert g.shape == (2, 8, 45, 80) and g.min() >= 0 and g.max() <= 1.0


def test_clip_shard_partition():
    from iip_uavsal_saliency_amd.parallel import ClipShard
    seen = []
    for r in range(8):
        s = ClipShard(64, 8, r)
        seen += list(range(s.first, s.first + s.count))
    assert seen == list(range(64))
    with pytest.raises(ValueError):
        ClipShard(10, 4, 0)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from iip_uavsal_saliency_amd.parallel import forward_clips_sharded
    from oracle.uavsal_ref import build_oracle
    model = build_oracle(time_dims=2)          # any module with forward_clips; the oracle runs on CPU
    C, T, H, W = 2, 2, 72, 104
    x = torch.from_numpy(synth.normalize_frames(synth.synth_frames_u8(C * T, H, W))).view(C, T, 3, H, W)
    cb = [torch.from_numpy(synth.gauss_priors(C * T, 9, 13)).view(C, T, 8, 9, 13),
          torch.from_numpy(synth.ob_priors(C * T, 9, 13)).view(C, T, 20, 9, 13)]
    out, st = forward_clips_sharded(model, x, cb)                       # full batch, sliced per rank
    # production form: the rank holds only its own shard of the frames
    out_l, st_l = forward_clips_sharded(model, x[rank:rank + 1], [cb[0][rank:rank + 1], cb[1][rank:rank + 1]],
                                        total_clips=C)
    assert torch.equal(out, out_l) and torch.equal(st, st_l)
    try:
        forward_clips_sharded(model, x, cb, total_clips=C)               # a full batch is not a shard
        raise AssertionError("shard size was not checked")
    except RuntimeError:
        pass
    q.put((rank, out.numpy(), st.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_forward_equals_single_process_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict()
    for _ in range(2):
        r, out, st = q.get(timeout=240)
        got[r] = (out, st)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from oracle.uavsal_ref import build_oracle
    model = build_oracle(time_dims=2)
    C, T, H, W = 2, 2, 72, 104
    x = torch.from_numpy(synth.normalize_frames(synth.synth_frames_u8(C * T, H, W))).view(C, T, 3, H, W)
    cb = [torch.from_numpy(synth.gauss_priors(C * T, 9, 13)).view(C, T, 8, 9, 13),
          torch.from_numpy(synth.ob_priors(C * T, 9, 13)).view(C, T, 20, 9, 13)]
    ref_out, ref_st = model.forward_clips(x, cb)
    for r in range(2):
        assert np.allclose(got[r][0], ref_out.numpy(), atol=1e-6)     # every rank holds all maps
        assert np.allclose(got[r][1], ref_st[r:r + 1].numpy(), atol=1e-6)   # states stay local
    assert np.array_equal(got[0][0], got[1][0])


def _clips(C, T, H, W):
    """C distinct clips (clip c seeded with c, bench.py's convention), so a gathered tensor in the wrong order cannot pass."""
    h, w = H // 8, W // 8
    xs = [torch.from_numpy(synth.normalize_frames(synth.synth_frames_u8(T, H, W, c))) for c in range(C)]
    g = [torch.from_numpy(synth.gauss_priors(T, h, w)) for _ in range(C)]
    o = [torch.from_numpy(synth.ob_priors(T, h, w, seed=c)) for c in range(C)]
    return torch.stack(xs), [torch.stack(g), torch.stack(o)]


def _worker_w4(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from iip_uavsal_saliency_amd.parallel import ClipShard, forward_clips_sharded
    from oracle.uavsal_ref import build_oracle
    model = build_oracle(time_dims=2)
    C, T, H, W = 8, 2, 72, 104
    x, cb = _clips(C, T, H, W)
    sh = ClipShard(C, world, rank)
    # production form: the rank hands over only its own two clips, total_clips names the batch
    out, st = forward_clips_sharded(model, sh.local(x).clone(), [sh.local(cb[0]).clone(), sh.local(cb[1]).clone()], total_clips=C)
    q.put((rank, sh.first, sh.count, out.numpy(), st.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_forward_gloo_world4_rank_order_is_clip_order():
    """W > 2: four ranks x two clips through `forward_clips_sharded(..., total_clips=8)` -- the gathered maps are in clip order
    on every rank (rank r's block at [2r, 2r + 2)), every clip differs from every other, states stay on their rank."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    W_ = 4
    procs = [ctx.Process(target=_worker_w4, args=(r, W_, port, q)) for r in range(W_)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(W_):
        r, first, count, out, st = q.get(timeout=300)
        got[r] = (first, count, out, st)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from oracle.uavsal_ref import build_oracle
    model = build_oracle(time_dims=2)
    x, cb = _clips(8, 2, 72, 104)
    ref_out, ref_st = model.forward_clips(x, cb)
    ref = ref_out.numpy()
    # the clips are told apart by more than the comparison tolerance (else the order check below would be vacuous)
    assert min(np.abs(ref[a] - ref[b]).max() for a in range(8) for b in range(a)) > 1e-3
    for r in range(W_):
        first, count, out, st = got[r]
        assert (first, count) == (2 * r, 2) and out.shape == ref.shape
        # (the ranks run one thread each, this process all of them: oneDNN's summation order differs by ~1e-6)
        assert np.abs(out - ref).max() < 1e-4                                     # all eight maps, clip order
        assert np.abs(st - ref_st[first:first + count].numpy()).max() < 1e-4      # only its own two states
        assert np.array_equal(out, got[0][2])                                     # bit-identical on every rank


def test_every_bias_type_builds_the_reference_module_tree():
    """reference model.py:281-324: a prior net exists iff its flag is set; the two fusion blocks iff any is; `fucb_layer` takes
    64 channels per enabled prior.  Same keys and shapes as the oracle tree for all eight flag sets (the [1,1,1] tree is pinned to
    the reference's 51.59 MB known answer; the oracle's other trees to reference goldens in tests/test_oracle_golden.py)."""
    import itertools
    from iip_uavsal_saliency_amd import UAVSal
    from oracle.uavsal_ref import RefUAVSal
    for bias in itertools.product((0, 1), repeat=3):
        m, r = UAVSal(bias_type=list(bias)), RefUAVSal(bias_type=bias)
        sd, rd = m.state_dict(), r.state_dict()
        assert list(sd.keys()) == list(rd.keys()) and all(sd[k].shape == rd[k].shape for k in sd), bias
        assert (m.use_gauss_prior, m.use_ob_prior, m.use_context_prior) == bias and m.num_cb == sum(bias)
        for attr, on in (("gauss_cb_layer", bias[0]), ("ob_cb_layer", bias[1]), ("cxt_cb_prior", bias[2]),
                         ("fucb_layer", any(bias)), ("fucbst_layer", any(bias))):
            assert hasattr(m, attr) == bool(on), (bias, attr)
        if any(bias):
            assert m.fucb_layer[0].conv[0][0].weight.shape[1] == 64 * sum(bias)
    with pytest.raises(ValueError):
        UAVSal(bias_type=[1, 2, 0])
    with pytest.raises(ValueError):
        UAVSal(bias_type=[1, 1])


def test_arena_planner_never_overlaps_live_buffers():
    """`engine.plan_arena`: buffers whose live ranges intersect get disjoint address ranges, the total stays near the
    largest simultaneously-live set, and disjoint lifetimes DO share (random instances + the degenerate ones)."""
    from iip_uavsal_saliency_amd.engine import plan_arena, ARENA_ALIGN
    rng = np.random.default_rng(7)
    rnd = lambda n: (n + ARENA_ALIGN - 1) // ARENA_ALIGN * ARENA_ALIGN
    for trial in range(40):
        n = int(rng.integers(1, 60))
        bufs = []
        for _ in range(n):
            f = int(rng.integers(0, 100))
            bufs.append((int(rng.integers(1, 5_000_000)), f, f + int(rng.integers(0, 30))))
        offs, total, bound = plan_arena(bufs)
        for i in range(n):
            assert offs[i] % ARENA_ALIGN == 0 and offs[i] + bufs[i][0] <= total
            for j in range(i):
                if not (bufs[i][2] < bufs[j][1] or bufs[j][2] < bufs[i][1]):       # live together
                    assert offs[i] + rnd(bufs[i][0]) <= offs[j] or offs[j] + rnd(bufs[j][0]) <= offs[i], (trial, i, j)
        assert bound <= total <= sum(rnd(b[0]) for b in bufs)
        assert total <= 1.6 * bound + ARENA_ALIGN, (trial, total, bound)          # (greedy-by-size: never far from the bound)
    # a chain of equal buffers, each live together with its neighbour only: two slots
    chain = [(1000, i, i + 1) for i in range(10)]
    offs, total, bound = plan_arena(chain)
    assert total == 2 * rnd(1000) == bound and len(set(offs)) == 2
    assert plan_arena([]) == ([], 0, 0)


def test_arena_layout_of_the_real_plans():
    """The sizing pass of the real launch plans on CPU (`plan_only`): every pair of activations that is live together is
    disjoint in the arena; a side lane's buffers are live from the fork to the join; the footprint is the live set, not the
    layer count -- 720x1280 4 x 16 frames (BASELINE configs[4]) fits 16 GB where one allocation per activation took 42.6."""
    from iip_uavsal_saliency_amd import UAVSal
    from iip_uavsal_saliency_amd.engine import Engine, ARENA_ALIGN, arena_conflict
    m = UAVSal(time_dims=8).eval()
    rnd = lambda n: (n + ARENA_ALIGN - 1) // ARENA_ALIGN * ARENA_ALIGN
    for (C, T, H, W, cap_mb) in ((1, 8, 360, 640, 500), (8, 8, 360, 640, 3600), (4, 16, 720, 1280, 12000)):
        e = Engine(m, "cpu", n_seq=C, seq_len=T, H=H, W=W, ctx_T=T, ctx_mode="clip", plan_only=True)
        lay = e.arena_layout()
        assert len(lay) == e.arena_stats["buffers"] > 80
        shared_on_a_lane = 0
        for i, ta in enumerate(lay):
            (a, oa, na, fa, la) = ta[:5]
            assert fa <= la and ta[6] <= ta[7]
            for tb in lay[:i]:
                (b, ob, nb, fb, lb) = tb[:5]
                disjoint = oa + rnd(na) <= ob or ob + rnd(nb) <= oa
                if arena_conflict(ta[2:], tb[2:]):
                    assert disjoint, (a, b)
                elif not disjoint and not (la < fb or lb < fa):
                    # same addresses while both are "live" on the main lane's clock: only two buffers private to ONE side lane
                    # between the same fork and join, used one after the other in that lane's stream order
                    assert isinstance(ta[5], tuple) and ta[5] == tb[5] and (ta[7] < tb[6] or tb[7] < ta[6]), (a, b)
                    shared_on_a_lane += 1
        assert shared_on_a_lane > 0          # (the two prior nets' expanded tensors on lane 1)
        st = e.arena_stats
        assert st["live_bound_mb"] <= st["arena_mb"] <= cap_mb and st["arena_mb"] < 0.3 * st["unshared_mb"], st
        # lanes: what the temporal branch of STBlock 0 touches on lane 6 is live from its fork to its join
        names = [o["name"] for o in e.ops_meta]
        if C * T > 8:        # (the temporal branch has its side lane from nine frames up: engine.py, UAVSAL_ST_LANES)
            fork = max(i for i, nm in enumerate(names) if nm == "fork6" and i < names.index("st0.sub.pw"))
            join = min(i for i, nm in enumerate(names) if nm == "join6" and i > names.index("st0.sub.pw"))
            te1 = [t for t in lay if t[0] == "st0_te1"][0]           # (written on lane 6, read on the main lane after the join)
            assert te1[3] <= fork and te1[4] >= join and te1[5] == "mixed"
        # the lateral convs of the SRF-Net head run on lane 6: x4 is written and read there only -- private to the lane
        fork = max(i for i, nm in enumerate(names) if nm == "fork6" and i < names.index("conv_lv4"))
        join = min(i for i, nm in enumerate(names) if nm == "join6" and i > names.index("conv_lv4"))
        x4 = [t for t in lay if t[0] == "x4"][0]
        assert x4[3] == fork and x4[4] == join and x4[5] == (6, fork) and fork < x4[6] <= x4[7] < join
    # taps keep their buffers to the end of the plan
    e = Engine(m, "cpu", n_seq=1, seq_len=4, H=96, W=160, ctx_T=4, ctx_mode="tile", plan_only=True, taps=True)
    last = len(e.ops_meta)
    for nm in ("sfnet", "st0", "st1", "prefuse", "rnn", "fu320", "f6", "f13", "f17"):
        assert [t for t in e.arena_layout() if t[0] == nm][0][4] == last, nm
    # every bias_type plans (no prior at all: no concat buffers)
    e0 = Engine(UAVSal(time_dims=4, bias_type=[0, 0, 0]).eval(), "cpu", n_seq=1, seq_len=4, H=96, W=160, ctx_T=4, ctx_mode="tile",
                plan_only=True)
    assert not any(t[0] in ("fu320", "cb192") for t in e0.arena_layout())


@pytest.mark.parametrize("bias", [(1, 1, 1), (0, 0, 0), (1, 0, 1), (0, 1, 0)])
def test_recording_pass_addresses_every_activation_inside_its_live_range(bias):
    """The engine's SECOND pass on the CPU (tests/mock_plan.py: real shape queries, stubbed plan recording): every activation
    address comes from the arena, which refuses one outside the buffer's declared live range -- so a recorder that forgets to
    declare a use fails here, at build time, on every plan variant (taps, persistent state, clips, split-fp16 shadows, frame-
    invariant priors, every bias_type).  Debug mode: one NaN fill per released buffer, on the main lane, right range."""
    import mock_plan
    from iip_uavsal_saliency_amd import UAVSal
    m = UAVSal(time_dims=4, bias_type=list(bias)).eval()
    small = dict(H=96, W=160, ctx_T=4)
    variants = [dict(n_seq=1, seq_len=4, ctx_mode="tile", **small), dict(n_seq=1, seq_len=8, ctx_mode="tile", taps=True, **small),
                dict(n_seq=2, seq_len=4, ctx_mode="clip", persistent=True, **small),
                dict(n_seq=4, seq_len=4, ctx_mode="clip", precision="f16x3", **small),
                dict(n_seq=1, seq_len=3, H=72, W=104, ctx_T=3, ctx_mode="clip", use_lanes=False)]
    if bias[0] or bias[1]:
        variants.append(dict(n_seq=1, seq_len=4, ctx_mode="clip", static_priors=True, **small))
    for kw in variants:
        for debug in (False, True):
            m.arena_debug = debug
            eng, mock = mock_plan.record(m, **kw)
            assert mock.n == len(eng.ops_meta) and eng._arena is not None
            fills = [o for o in eng.ops_meta if o["kind"] == "poison"]
            if not debug:
                assert not fills and not mock.fills
                continue
            by_range = {(eng._arena.data_ptr() + 4 * t[1], t[2]): t[0] for t in eng.arena_layout()}
            logical = len(eng.ops_meta) - len(fills)
            released = [t for t in eng.arena_layout() if (t[7] if isinstance(t[5], tuple) else t[4]) < logical - 1]
            assert len(mock.fills) == len(fills) == len(released), (kw, len(fills), len(released))
            lanes_of = {}         # (two buffers of equal size may share one range: one after the other on a lane, or on the main lane)
            for aid, off, n_, _, _, lk, _, _ in eng.arena_layout():
                lanes_of.setdefault((eng._arena.data_ptr() + 4 * off, n_), {0}).add(lk[0] if isinstance(lk, tuple) else 0)
            for _, ptr, n, lane in mock.fills:
                assert (ptr, n) in by_range and lane in lanes_of[(ptr, n)], (kw, lane)
            # a fill sits behind the last op that may touch its buffer: `last` counts logical ops, fills excluded
            pos, k = {}, 0
            for o in eng.ops_meta:
                if o["kind"] == "poison":
                    pos[o["name"][len("poison:"):]] = k
                else:
                    k += 1
            for aid, _, _, first, last, lkey, lfirst, llast in released:
                # (a buffer private to a side lane is released in that lane's order, by a fill on that lane)
                assert pos[str(aid)] == (llast if isinstance(lkey, tuple) else last) + 1, (aid, first, last, lkey, pos[str(aid)])


def test_reference_style_whole_model_pickle_loads_through_the_shim(tmp_path):
    """A whole pickled model whose classes live in `model`, `model_feature`, `model_convlstm` and
    `torchvision.models.mobilenet` (as the reference's checkpoints do, Demo_Train_Test.py:159-160)
    loads without those modules being importable."""
    import sys
    import types
    from iip_uavsal_saliency_amd import UAVSal, model as M, model_feature as MF, model_convlstm as MC
    from iip_uavsal_saliency_amd.checkpoint import load_reference_state_dict, load_reference_checkpoint

    src = UAVSal(time_dims=3)
    synth.load_synth_weights(src, 1)
    src._engines = {}
    renames = {M.UAVSal: ("model", "UAVSal"), M.dwBlock: ("model", "dwBlock"), M.BasicConv2d: ("model", "BasicConv2d"),
               M.STBlock: ("model", "STBlock"), M.spConv: ("model", "spConv"), M.teConv_sub: ("model", "teConv_sub"),
               M.uavsal_srfnet_aspp: ("model", "uavsal_srfnet_aspp"),
               MC.ConvTWA: ("model_convlstm", "ConvTWA"), MC.ConvTWACell: ("model_convlstm", "ConvTWACell"),
               MF.ReMobileNetV2: ("model_feature", "ReMobileNetV2"),
               MF.ConvBNReLU: ("torchvision.models.mobilenet", "ConvBNReLU"),
               MF.InvertedResidual: ("torchvision.models.mobilenet", "InvertedResidual")}
    saved = {c: (c.__module__, c.__qualname__, c.__name__) for c in renames}
    fakes = {}
    try:
        for cls, (mod, name) in renames.items():
            parts = mod.split(".")
            for i in range(1, len(parts) + 1):
                fakes.setdefault(".".join(parts[:i]), types.ModuleType(".".join(parts[:i])))
            setattr(fakes[mod], name, cls)
            cls.__module__, cls.__qualname__, cls.__name__ = mod, name, name
        sys.modules.update(fakes)
        path = str(tmp_path / "uavsal-mobilenet_v2-fake.pth")
        torch.save(src, path)                       # whole-model pickle, reference style
    finally:
        for cls, (m, q, n) in saved.items():
            cls.__module__, cls.__qualname__, cls.__name__ = m, q, n
        for k in fakes:
            sys.modules.pop(k, None)
    assert "model" not in sys.modules and "torchvision" not in sys.modules
    sd = load_reference_state_dict(path)
    assert len(sd) == 685
    dst = UAVSal(time_dims=3)
    load_reference_checkpoint(dst, path)
    for k, v in src.state_dict().items():
        assert torch.equal(v, dst.state_dict()[k]), k
    # the constructor's pre_model_path (reference model.py:338-339) goes through the same shim
    dst2 = UAVSal(time_dims=3, pre_model_path=path)
    for k, v in src.state_dict().items():
        assert torch.equal(v, dst2.state_dict()[k]), k


def test_mat_reader_and_priors(golden_dir):
    """The minimal HDF5 reader against the reference's own prior files (when present) and the
    converted fixtures; the closed-form gaussian priors equal the shipped gauss_priors.mat exactly."""
    from iip_uavsal_saliency_amd import matio, priors
    g_fix = np.load(os.path.join(golden_dir, "gauss_priors.npz"))["PriorMaps"]
    closed = synth.gauss_priors(1, 45, 80)[0].transpose(1, 2, 0)
    assert g_fix.shape == (45, 80, 8) and np.array_equal(closed, g_fix)
    ref_dir = "/root/reference"
    if os.path.isdir(ref_dir):
        for name in ("gauss_priors", "UAV2_ob_priors_train", "AVS1K_ob_priors_train"):
            a = matio.loadmat(os.path.join(ref_dir, name + ".mat"))["PriorMaps"]
            assert np.array_equal(a, np.load(os.path.join(golden_dir, name + ".npz"))["PriorMaps"])
    cb = priors.get_bias([1, 1, 1], 3, 45, 80, ob_prior_path=os.path.join(golden_dir, "UAV2_ob_priors_train.npz"),
                         device="cpu")
    assert tuple(cb[0].shape) == (3, 8, 45, 80) and tuple(cb[1].shape) == (3, 20, 45, 80)
    assert cb[0].dtype == torch.float32 and float(cb[1].max()) <= 1.0
    # ... as a zero-stride view of ONE map set (what lets the model run its prior nets once per call); same values materialised
    from iip_uavsal_saliency_amd import UAVSal
    rep = priors.get_bias([1, 1, 1], 3, 45, 80, ob_prior_path=os.path.join(golden_dir, "UAV2_ob_priors_train.npz"),
                          device="cpu", broadcast=False)
    assert cb[0].stride(0) == 0 and cb[1].stride(0) == 0 and rep[0].stride(0) != 0
    assert torch.equal(cb[0], rep[0]) and torch.equal(cb[1], rep[1])
    assert UAVSal.frame_invariant(cb, 1) and not UAVSal.frame_invariant(rep, 1)
    assert not UAVSal.frame_invariant([cb[0], rep[1]], 1)                       # both tensors must be broadcasts
    assert not UAVSal.frame_invariant([c.contiguous() for c in cb], 1)          # never decided from the values
    five = [c[None].expand(2, -1, -1, -1, -1) for c in cb]                      # forward_clips: [C, T, ., h, w]
    assert UAVSal.frame_invariant(five, 2)
    per_clip = [torch.stack([c[0], c[0] + 1])[:, None].expand(-1, 3, -1, -1, -1) for c in cb]     # one map set PER CLIP: general plan
    assert not UAVSal.frame_invariant(per_clip, 2)


def test_priors_resize_path_letterbox_geometry_and_uint8_truncation(golden_dir):
    """priors stored at another size are letterboxed by `padding()` into a uint8 array
    (reference utils_data.py:321-343, 460-464, 595-599), which truncates the [0,1] floats to {0,1}."""
    from iip_uavsal_saliency_amd import priors
    # hand-computed: 4x6 -> 2x3 is a scale of exactly 2, so cv2's half-pixel rule samples at 0.5, 2.5, ...: every output
    # is the mean of a 2x2 block.  Blocks: all ones -> 1.0 (survives the uint8 cast), three ones + 0 -> 0.75 (truncated
    # to 0), 0.5s -> 0.5 (0)
    m = np.array([[1, 1, 1, 1, .5, .5],
                  [1, 1, 1, 0, .5, .5],
                  [0, 0, 1, 1, 1, 1],
                  [0, 0, 1, 1, 1, 1]], np.float32)
    f = priors.letterbox(m, 2, 3, quirk=False)
    assert f.dtype == np.float32 and np.array_equal(f, np.array([[1, .75, .5], [0, 1, 1]], np.float32))
    q = priors.letterbox(m, 2, 3)
    assert q.dtype == np.uint8 and np.array_equal(q, np.array([[1, 0, 0], [0, 1, 1]], np.uint8))
    # geometry, both branches of padding(): rows_rate <= cols_rate pastes full-width rows in the middle ...
    ones = np.ones((45, 80), np.float32)
    a = priors.letterbox(ones, 40, 64)                  # new_rows = 45 * 64 // 80 = 36, rows 2..37
    assert a[2:38].min() == 1 and a[:2].max() == 0 and a[38:].max() == 0
    b = priors.letterbox(ones, 36, 80)                  # rows_rate 1.25 > cols_rate 1: new_cols = 80 * 36 // 45 = 64, cols 8..71
    assert b[:, 8:72].min() == 1 and b[:, :8].max() == 0 and b[:, 72:].max() == 0
    # ... and the sizes BASELINE configs 0 and 4 need from the shipped 45x80 files: 36x64 and 90x160 (same aspect: no bars)
    path = os.path.join(golden_dir, "UAV2_ob_priors_train.npz")
    stored = np.load(path)["PriorMaps"]
    for (r, c) in ((36, 64), (90, 160)):
        cb = priors.get_bias([1, 1, 1], 2, r, c, ob_prior_path=path, gauss_prior_path=os.path.join(golden_dir, "gauss_priors.npz"),
                             device="cpu")
        assert tuple(cb[0].shape) == (2, 8, r, c) and tuple(cb[1].shape) == (2, 20, r, c) and cb[1].dtype == torch.float32
        assert set(np.unique(cb[0].numpy())) <= {0.0, 1.0} and set(np.unique(cb[1].numpy())) <= {0.0, 1.0}     # the quirk
        fl = priors.get_ob_priors(path, 1, r, c, quirk=False)[0]
        assert fl.dtype == np.float32 and 0.0 <= fl.min() and fl.max() <= stored.max() + 1e-6
        assert np.array_equal(priors.get_ob_priors(path, 1, r, c)[0], fl.astype(np.uint8))
        assert abs(float(fl.mean()) - float(stored.mean())) < 0.02 * float(stored.mean())      # a resize, not a crop
    # without a gaussian file the reference computes the closed form AT the requested size (utils_data.py:452-456)
    g = priors.get_guasspriors(1, 36, 64)
    assert g.dtype == np.float32 and g.shape == (1, 36, 64, 8) and np.array_equal(g[0], synth.gauss_priors(1, 36, 64)[0].transpose(1, 2, 0))


# ---- bench.py as its own launcher -----------------------------------------------
_STUB_RANK = r'''
import json, os, sys
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["LOCAL_RANK"] == os.environ["RANK"] and os.environ["MASTER_ADDR"] == "127.0.0.1"
assert int(os.environ["MASTER_PORT"]) > 0 and os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
open(os.path.join(sys.argv[1], "rank%d.json" % rank), "w").write(json.dumps(
    {"rank": rank, "world": world, "port": os.environ["MASTER_PORT"], "argv": sys.argv[2:]}))
if os.environ.get("STUB_FAIL_RANK") == str(rank):
    sys.exit(7)
if os.environ.get("STUB_FAIL_RANK") is not None:
    import time
    time.sleep(120)              # a rank stuck in a collective: the launcher must stop it
if rank == 0:
    print(json.dumps({"metric": "stub", "n_gpus": world}), flush=True)
'''


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("uavsal_bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bench_self_launch_starts_n_fresh_ranks(tmp_path, monkeypatch, capfd):
    bench = _bench_module()
    stub = tmp_path / "stub_rank.py"
    stub.write_text(_STUB_RANK)
    monkeypatch.setenv("UAVSAL_BENCH_VISIBLE_GPUS", "2")
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    rc = bench.self_launch(2, [], worker=[sys.executable, str(stub), str(tmp_path), "--gpus", "2"])
    assert rc == 0
    recs = [json.loads((tmp_path / ("rank%d.json" % r)).read_text()) for r in range(2)]
    assert [r["rank"] for r in recs] == [0, 1] and all(r["world"] == 2 for r in recs)
    assert recs[0]["port"] == recs[1]["port"] and recs[0]["argv"] == ["--gpus", "2"]
    out = capfd.readouterr().out.strip().splitlines()
    assert len(out) == 1 and json.loads(out[0]) == {"metric": "stub", "n_gpus": 2}      # rank 0's line, once


def test_bench_self_launch_fails_loudly(tmp_path, monkeypatch, capfd):
    bench = _bench_module()
    stub = tmp_path / "stub_rank.py"
    stub.write_text(_STUB_RANK)
    # (a) a failing rank: its code is the launcher's, the rank left waiting is stopped (the call returns long before 120 s)
    monkeypatch.setenv("UAVSAL_BENCH_VISIBLE_GPUS", "2")
    monkeypatch.setenv("STUB_FAIL_RANK", "1")
    import time
    t0 = time.perf_counter()
    rc = bench.self_launch(2, [], worker=[sys.executable, str(stub), str(tmp_path)])
    assert rc == 7 and time.perf_counter() - t0 < 60
    assert "rank 1 exited with code 7" in capfd.readouterr().err
    # (b) fewer GPUs than asked for: refuse, start nothing
    monkeypatch.delenv("STUB_FAIL_RANK")
    monkeypatch.setenv("UAVSAL_BENCH_VISIBLE_GPUS", "1")
    for f in tmp_path.glob("rank*.json"):
        f.unlink()
    assert bench.self_launch(2, [], worker=[sys.executable, str(stub), str(tmp_path)]) == 2
    assert "only 1 GPU(s) visible" in capfd.readouterr().err and not list(tmp_path.glob("rank*.json"))
    # (c) every rank hangs (a collective nobody leaves, a hung GPU): the wall-clock limit stops them by PID and says which
    monkeypatch.setenv("UAVSAL_BENCH_VISIBLE_GPUS", "2")
    monkeypatch.setenv("STUB_FAIL_RANK", "99")            # nobody fails, everybody sleeps 120 s
    t0 = time.perf_counter()
    rc = bench.self_launch(2, [], worker=[sys.executable, str(stub), str(tmp_path)], deadline_s=3.0)
    assert rc == 124 and time.perf_counter() - t0 < 40
    assert "still running and have been stopped" in capfd.readouterr().err


def test_bench_counts_gpus_without_a_hip_call(monkeypatch):
    """The launcher's GPU count comes from the visibility variables or the KFD topology, not from the HIP runtime."""
    bench = _bench_module()
    monkeypatch.delenv("UAVSAL_BENCH_VISIBLE_GPUS", raising=False)
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: (_ for _ in ()).throw(AssertionError("HIP was asked")))
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1,2")
    assert bench.visible_gpus() == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert bench.visible_gpus() == 0
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    for var in ("CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    if os.path.isdir("/sys/class/kfd/kfd/topology/nodes"):
        assert bench.visible_gpus() >= 0                 # read from sysfs: still no HIP call


def test_bench_gpus_2_on_a_box_without_two_gpus_exits_nonzero():
    """The command shape the driver uses (`python bench.py --gpus N`, no launcher around it) must not silently measure one
    GPU: in this container (no GPU) and on a 1-GPU box it exits non-zero with a message and prints no JSON line."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["UAVSAL_BENCH_VISIBLE_GPUS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "only 1 GPU(s) visible" in r.stderr and r.stdout.strip() == ""
    # under a launcher whose world size differs from --gpus it refuses too
    env["WORLD_SIZE"], env["RANK"], env["LOCAL_RANK"] = "1", "0", "0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and r.stdout.strip() == ""


def test_committed_bench_line_carries_the_contract_fields():
    """The bench line the round's last collection produced (profiles/r5_bench_default.json, written by `python bench.py` on the
    MI355X): every field of the driver's contract, the roofline and cpu_baseline objects, and a stamp-matched PMC traffic figure."""
    import json
    r = json.load(open(os.path.join(ROOT, "profiles", "r5_bench_default.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "parity", "ranks_seen", "scaling_reference"):
        assert k in r, k
    assert r["n_gpus"] == 1 and r["higher_is_better"] is True and r["scaling"] == "weak" and r["vs_baseline"] is None
    assert r["dtype"] == "f32" and r["data"] == "synthetic" and "workload" in r["config"] and "model" not in r["config"]
    assert abs(r["value"] - 8 * r["steps"] / (r["ms_per_step"] * 1e-3 * r["steps"])) < 1.0          # frames / time of the median window
    ro = r["roofline"]
    assert ro["bound"] in ("hbm", "mfma") and ro["unit"] in ("GB/s", "TFLOP/s") and abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-3
    assert ro["traffic"] is not None and ro["in_loop"]["status"] == "ok"
    cb = r["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb and cb["unit"] == "frames/s"
    assert r["parity"]["max_abs_map_vs_cpu_ref"] <= r["parity"]["tolerance"] == 1e-3
    fam = r["scaling_reference"]["roofline_dw"]
    assert fam["frac"] >= 0.60          # north_star: >= 60 % of the HBM roofline on the depthwise convs
    # ... and not as an artefact of timing back-to-back repeats: the same launches in a --lanes 0 kernel trace of the whole step
    assert fam["in_loop_lanes_off"]["status"] == "ok" and fam["in_loop_lanes_off"]["frac"] >= 0.60
    assert r["roofline_dw"]["in_loop"]["status"] == "ok" and "per_op_timing" in r
    # the reference's own shapes are timed and checked in the same line
    for k in ("extra_demo_default", "extra_288x512"):
        assert r[k]["value"] > 0 and r[k]["max_abs_map_vs_cpu_ref"] <= 1e-3 and r[k]["cpu_oracle_frames_per_s"] > 0 and r[k]["first_call_ms"] > 0
    assert r["first_call_ms"] > 0 and r["activation_arena"]["arena_mb"] < 0.25 * r["activation_arena"]["unshared_mb"]
    assert len(r["scaling_reference"]["windows_ms_per_step"]) == 3 and r["scaling_reference"]["steps"] == r["steps"]


def test_model_copies_carry_parameters_and_settings_never_runtime_handles():
    """`torch.save(model)` (how the reference stores its checkpoints, model.py:339), pickle and deepcopy: parameters and settings
    travel, launch plans / packed device weights / streams (process-local handles) do not; `replica()` shares the packed weights."""
    import copy
    import ctypes
    import io
    from iip_uavsal_saliency_amd import UAVSal, UAVSAL_LSTM, synth
    for cls in (UAVSal, UAVSAL_LSTM):
        m = cls(time_dims=4)
        synth.load_synth_weights(m, 0)
        m.eval()
        m._engines["k"] = ctypes.c_void_p(1234)              # what a recorded plan holds: not picklable, not copyable
        m._wshared["cuda:0"] = {"w": ctypes.c_void_p(5)}
        m.__dict__["_stream_replicas"] = [m.replica()]
        m.precision, m.persistent_state = "f16x3", True
        buf = io.BytesIO()
        torch.save(m, buf)
        buf.seek(0)
        for d in (copy.deepcopy(m), torch.load(buf, weights_only=False)):
            assert type(d) is cls and not d.training
            assert len(d._engines) == 0 and d._wshared == {} and d._wversion is None and "_stream_replicas" not in d.__dict__
            assert d.precision == "f16x3" and d.persistent_state is True
            sd, sm = d.state_dict(), m.state_dict()
            assert list(sd) == list(sm) and all(torch.equal(sd[k], sm[k]) for k in sm)
            assert next(d.parameters()).data_ptr() != next(m.parameters()).data_ptr()
        r = m.replica()
        assert r._wshared is m._wshared and len(r._engines) == 0
        assert next(r.parameters()).data_ptr() == next(m.parameters()).data_ptr()

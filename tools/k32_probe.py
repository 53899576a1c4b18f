"""fp32 GEMM with 32-float (full cache line) K stages -- tiles 8 / 9 of uavsal_conv_gemm -- against the 16-float
instances (tiles 1 / 7): parity vs F.conv2d on the CPU, then hipEvent timings on the path's shapes.
  python tools/k32_probe.py parity      parity only
  python tools/k32_probe.py time        timings only
  python tools/k32_probe.py pmc TILE    a few launches of one instance (for `rocprofv3 --pmc ...`)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F
from iip_uavsal_saliency_amd import _lib as L, ops, packing as P

lib = L.load()
dev = torch.device("cuda")


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


def parity():
    cases = [  # n, h, w, cin, cout, k, act, res
        (2, 12, 20, 256, 1536, 1, 1, False), (1, 23, 40, 1536, 256, 1, 0, True), (3, 7, 5, 320, 256, 1, 1, False),
        (1, 45, 80, 32, 256, 1, 1, True), (2, 45, 80, 64, 384, 1, 1, False), (2, 13, 17, 96, 200, 1, 0, False),
        (1, 12, 20, 64, 96, 3, 1, False), (2, 9, 13, 448, 256, 3, 1, False), (1, 45, 80, 256, 256, 3, 0, True),
        (8, 45, 80, 256, 1536, 1, 1, False)]
    bad = 0
    for (n, h, w, cin, cout, k, act, use_res) in cases:
        x = rnd((n, cin, h, w), 1, 2.0)
        wt = rnd((cout, cin, k, k), 2, 1.0 / np.sqrt(cin * k * k))
        scale = rnd((cout,), 3) * 0.5 + 1.0
        bias = rnd((cout,), 4)
        res = rnd((n, cout, h, w), 5) if use_res else None
        y = F.conv2d(x, wt, padding=k // 2) * scale.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)
        ref = torch.clamp(y, 0, 6) if act == 1 else y
        if use_res:
            ref = ref + res
        xg = x.permute(0, 2, 3, 1).contiguous().cuda()
        rg = res.permute(0, 2, 3, 1).contiguous().cuda() if use_res else None
        for tile in (1, 8, 9, 10):
            got = ops.conv_gemm(xg, wt, scale, bias, act=act, res=rg, prec="f32", tile=tile)
            err = (got.permute(0, 3, 1, 2).cpu() - ref).abs().max().item()
            ok = err <= 8e-5
            bad += not ok
            print("parity n=%d %dx%d cin=%d cout=%d k=%d act=%d res=%d tile=%d: max-abs %.2e %s" % (
                n, h, w, cin, cout, k, act, use_res, tile, err, "ok" if ok else "FAIL"), flush=True)
    print("parity: %d failures" % bad, flush=True)
    return bad


def run(hw, n_img, cin, cout, taps, tile, iters=20, launches_only=0, act=1, stamps=False):
    h, w = hw
    a = torch.rand((n_img * h * w, cin), device=dev) * 2 - 1
    wt = (torch.rand((cout, cin, 3 if taps == 9 else 1, 3 if taps == 9 else 1)) - 0.5) * 0.1
    wp = P.pack_conv_weight(wt, "f32k32" if tile in (8, 9, 10, 11) else "f32").to(dev)
    out = torch.empty((n_img * h * w, cout), device=dev)
    s = torch.ones(P.roundup(cout, 32), device=dev)
    b = torch.zeros(P.roundup(cout, 32), device=dev)
    d = L.ConvDesc()
    d.a, d.lda, d.a_img_stride = a.data_ptr(), cin, h * w
    d.w = wp.data_ptr()
    d.scale, d.bias = s.data_ptr(), b.data_ptr()
    d.out, d.ldc, d.o_img_stride = out.data_ptr(), cout, h * w
    d.n_img, d.H, d.W, d.Cin, d.Cout, d.taps = n_img, h, w, cin, cout, taps
    d.prec, d.act, d.epi, d.tile = L.PREC["f32"], act, 0, tile
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    if os.environ.get("PROBE_WS"):       # give the launch the K-split / stream-K workspace
        ws = torch.zeros(int(lib.uavsal_streamk_workspace_bytes()), dtype=torch.uint8, device=dev)
        d.sk_ws, d.sk_ws_bytes = ws.data_ptr(), ws.numel()
    if stamps:      # libuavsal_hip_stamps.so: the kernel adds its cycle sums into the K-split area of the workspace
        ws = torch.zeros(1 << 20, dtype=torch.uint8, device=dev)
        d.sk_ws, d.sk_ws_bytes = ws.data_ptr(), ws.numel()
        for _ in range(3):
            L.check(lib.uavsal_conv_gemm(C.byref(d), st), "conv")
        torch.cuda.synchronize()
        ws.zero_()
        L.check(lib.uavsal_conv_gemm(C.byref(d), st), "conv")
        torch.cuda.synchronize()
        v = ws[65536:65536 + 56].view(torch.int64).cpu().tolist()
        return v
    if launches_only:
        for _ in range(launches_only):
            L.check(lib.uavsal_conv_gemm(C.byref(d), st), "conv")
        torch.cuda.synchronize()
        return 0.0, 0.0
    plan = C.c_void_p(lib.uavsal_plan_create())
    lib.uavsal_plan_add_conv(plan, C.byref(d))
    ms = C.c_float()
    lib.uavsal_plan_time(plan, 0, 1, 3, st, C.byref(ms))
    L.check(lib.uavsal_plan_time(plan, 0, 1, iters, st, C.byref(ms)), "time")
    lib.uavsal_plan_destroy(plan)
    fl = 2.0 * n_img * h * w * cin * cout * taps
    return ms.value, fl / ms.value / 1e9


SHAPES = [((45, 80), 8, 256, 1536, 1), ((45, 80), 8, 320, 1920, 1), ((45, 80), 8, 64, 384, 1),
          ((45, 80), 8, 192, 1152, 1), ((45, 80), 8, 256, 256, 1), ((45, 80), 8, 1536, 256, 1),
          ((45, 80), 8, 448, 256, 9), ((45, 80), 64, 256, 1536, 1), ((45, 80), 64, 1536, 256, 1),
          ((45, 80), 8, 4096, 1536, 1)]


def timing():
    for sh in SHAPES:
        line = []
        for tile in (1, 7, 8, 9, 10):
            ms, tf = run(*sh, tile)
            line.append("t%d %7.1f us %6.1f TF" % (tile, ms * 1e3, tf))
        print("hw=%s n=%d K=%d N=%d taps=%d : %s" % (sh[0], sh[1], sh[2], sh[3], sh[4], " | ".join(line)), flush=True)


def parts():
    """UAVSAL_HIP_LIB=tools/_tmp/libuavsal_hip_probe.so: the kernel with parts compiled out."""
    names = {1: "full", 101: "no store", 116: "no epilogue", 102: "no MFMA", 104: "no DMA", 108: "no frag reads",
             106: "no MFMA, no DMA", 110: "no MFMA, no frag", 126: "K loop: DMA only", 124: "K loop: MFMA+frag only (no DMA, no epi)"}
    for sh in (((45, 80), 8, 256, 1536, 1), ((45, 80), 64, 256, 1536, 1), ((45, 80), 8, 4096, 1536, 1)):
        for tile in (8, 9, 10, 11):
            for act, nm in names.items():
                ms, tf = run(*sh, tile, act=act)
                print("parts n=%d K=%d N=%d tile=%d %-42s %8.1f us" % (sh[1], sh[2], sh[3], tile, nm, ms * 1e3), flush=True)


def stamps():
    """UAVSAL_HIP_LIB=tools/_tmp/libuavsal_hip_stamps.so: where the waves spend their cycles."""
    for sh in (((45, 80), 8, 256, 1536, 1), ((45, 80), 64, 256, 1536, 1), ((45, 80), 8, 4096, 1536, 1), ((45, 80), 8, 448, 256, 9)):
        for tile in (8, 10):
            v = run(*sh, tile, stamps=True)
            tot = max(v[4], 1)
            if tile == 10:
                print("stamps n=%d K=%d N=%d taps=%d tile=10 waves=%d: %.0f shader cycles per wave, in-kernel clock %.3f GHz" % (
                    sh[1], sh[2], sh[3], sh[4], v[5], tot / max(v[5], 1), 0.1 * tot / max(v[6], 1)), flush=True)
                continue
            print("stamps n=%d K=%d N=%d taps=%d tile=%d waves=%d: wait %.1f%% body %.1f%% tail %.1f%% epilogue %.1f%% | per wave %.0f cycles" % (
                sh[1], sh[2], sh[3], sh[4], tile, v[5], 100.0 * v[0] / tot, 100.0 * v[1] / tot, 100.0 * v[2] / tot,
                100.0 * v[3] / tot, tot / max(v[5], 1)), flush=True)


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "all"
    if mode == "parts":
        parts(); sys.exit(0)
    if mode == "stamps":
        stamps(); sys.exit(0)
    if mode == "time8":      # tile 8 only, the shapes that matter (variant A/B through UAVSAL_HIP_LIB)
        tl = int(sys.argv[2]) if len(sys.argv) > 2 else 8
        for sh in (SHAPES[0], SHAPES[1], SHAPES[5], SHAPES[6], SHAPES[7], SHAPES[9]):
            ms, tf = run(*sh, tl)
            print("n=%d K=%d N=%d taps=%d tile=%d: %8.1f us %6.1f TF" % (sh[1], sh[2], sh[3], sh[4], tl, ms * 1e3, tf), flush=True)
        sys.exit(0)
    if mode == "small":      # few tiles, long K: the backbone-tail projections, 64 x 64 instances (4 = 16-float stages)
        for sh in (((12, 20), 8, 960, 160, 1), ((12, 20), 8, 960, 320, 1), ((12, 20), 8, 1920, 256, 1), ((12, 20), 8, 1024, 256, 1),
                   ((23, 40), 8, 576, 96, 1), ((23, 40), 8, 384, 64, 1), ((23, 40), 1, 1536, 64, 1), ((12, 20), 8, 160, 960, 1)):
            line = []
            for tile in (0, 4, 8, 11):
                ms, tf = run(*sh, tile)
                line.append("t%d %6.1f us" % (tile, ms * 1e3))
            print("hw=%s n=%d K=%d N=%d : %s" % (sh[0], sh[1], sh[2], sh[3], " | ".join(line)), flush=True)
        sys.exit(0)
    if mode == "pmc":
        tile = int(sys.argv[2])
        run((45, 80), 64, 256, 1536, 1, tile, launches_only=4)
        sys.exit(0)
    rc = 0
    if mode in ("all", "parity"):
        rc = parity()
    if mode in ("all", "time"):
        timing()
    sys.exit(1 if rc else 0)

"""Probe of the pre-split (LDS-DMA) f16x3 GEMM: tile choice, with / without the output store.
The no-store / parts variants need the library built with UAVSAL_EXTRA_HIPCC_FLAGS=-DUAVSAL_PROBE
(python -m iip_uavsal_saliency_amd.build --force); on the product build they time the full kernel."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iip_uavsal_saliency_amd import _lib as L, packing as P, ops

lib = L.load()
dev = torch.device("cuda")


def run(hw, n_img, cin, cout, taps, tile, split=True, act=1, iters=20, prec="f16x3"):
    h, w = hw
    a = torch.rand((n_img * h * w, cin), device=dev) * 2 - 1
    wt = (torch.rand((cout, cin, 3 if taps == 9 else 1, 3 if taps == 9 else 1)) - 0.5) * 0.1
    out = torch.empty((n_img * h * w, cout), device=dev)
    s = torch.ones(P.roundup(cout, 32), device=dev)
    b = torch.zeros(P.roundup(cout, 32), device=dev)
    d = L.ConvDesc()
    d.a, d.lda, d.a_img_stride = a.data_ptr(), cin, h * w
    if split:
        sp = ops.split_shadow(a)
        d.a_split, d.ldas = sp.data_ptr(), 2 * cin
    d.scale, d.bias = s.data_ptr(), b.data_ptr()
    d.out, d.ldc, d.o_img_stride = out.data_ptr(), cout, h * w
    d.n_img, d.H, d.W, d.Cin, d.Cout, d.taps = n_img, h, w, cin, cout, taps
    d.prec, d.act, d.epi, d.tile = L.PREC[prec], act, 0, tile
    d.w = 1 << 20
    uses = int(lib.uavsal_conv_uses_split(C.byref(d))) == 1
    wp = P.pack_conv_weight(wt, "f16x3i" if uses else prec).to(dev)
    d.w = wp.data_ptr()
    plan = C.c_void_p(lib.uavsal_plan_create())
    lib.uavsal_plan_add_conv(plan, C.byref(d))
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ms = C.c_float()
    L.check(lib.uavsal_plan_time(plan, 0, 1, 3, st, C.byref(ms)), "time")
    L.check(lib.uavsal_plan_time(plan, 0, 1, iters, st, C.byref(ms)), "time")
    lib.uavsal_plan_destroy(plan)
    fl = 2.0 * n_img * h * w * cin * cout * taps
    return ms.value * 1e3, fl / ms.value / 1e9, uses


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":       # one configuration, for rocprofv3 --pmc
        n, K, N, taps, tile, split, act = (int(v) for v in sys.argv[2:9])
        us, tf, uses = run((45, 80), n, K, N, taps, tile, bool(split), act, iters=5)
        print("n=%d K=%d N=%d taps=%d tile=%d split=%d act=%d: %.1f us %.1f TF-eq" % (n, K, N, taps, tile, uses, act, us, tf))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "parts":     # library built with -DUAVSAL_PROBE: what bounds the K loop
        for sh in [((45, 80), 64, 1536, 256, 1), ((45, 80), 64, 256, 1536, 1)]:
            for tile in (1, 5, 6):
                for act, what in ((100, "all, no store"), (101, "DMA + barriers only"), (102, "MFMA + LDS reads only")):
                    us, tf, uses = run(*sh, tile, True, act)
                    print("K=%d N=%d tile=%d %-24s: %8.1f us (%6.1f TF-eq if it were the whole kernel)" % (sh[2], sh[3], tile, what, us, tf), flush=True)
        sys.exit(0)
    shapes = [((45, 80), 8, 256, 1536, 1), ((45, 80), 64, 256, 1536, 1), ((45, 80), 8, 1536, 256, 1),
              ((45, 80), 64, 1536, 256, 1), ((45, 80), 8, 448, 256, 9)]
    for sh in shapes:
        for tile in (1, 5, 6):
            for split in (True, False):
                for act in (1, 100):
                    if act == 100 and not split:
                        continue
                    us, tf, uses = run(*sh, tile, split, act)
                    print("hw=%s n=%d K=%d N=%d taps=%d tile=%d %s %s: %8.1f us %7.1f TF-eq" % (
                        sh[0], sh[1], sh[2], sh[3], sh[4], tile, "split" if uses else "regst",
                        "nostore" if act == 100 else "store  ", us, tf), flush=True)

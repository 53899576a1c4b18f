"""Can an HBM-bound depthwise launch run UNDER an MFMA-bound GEMM launch on another stream?
Times GEMM alone, depthwise alone, and both launched back to back on two streams (hipEvents, 20 repetitions)."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iip_uavsal_saliency_amd import _lib as L, packing as P

lib = L.load()
dev = torch.device("cuda")


def gemm_plan(n_img, cin, cout, prec, tile, hw=(45, 80)):
    h, w = hw
    a = torch.rand((n_img * h * w, cin), device=dev) * 2 - 1
    wt = (torch.rand((cout, cin, 1, 1)) - 0.5) * 0.1
    out = torch.empty((n_img * h * w, cout), device=dev)
    s = torch.ones(P.roundup(cout, 32), device=dev)
    b = torch.zeros(P.roundup(cout, 32), device=dev)
    wp = P.pack_conv_weight(wt, prec).to(dev)
    d = L.ConvDesc()
    d.a, d.lda, d.a_img_stride = a.data_ptr(), cin, h * w
    d.scale, d.bias = s.data_ptr(), b.data_ptr()
    d.out, d.ldc, d.o_img_stride = out.data_ptr(), cout, h * w
    d.n_img, d.H, d.W, d.Cin, d.Cout, d.taps = n_img, h, w, cin, cout, 1
    d.prec, d.act, d.epi, d.tile = L.PREC[prec], 1, 0, tile
    d.w = wp.data_ptr()
    plan = C.c_void_p(lib.uavsal_plan_create())
    lib.uavsal_plan_add_conv(plan, C.byref(d))
    return plan, (a, out, s, b, wp)


def dw_plan(n_img, c, hw=(45, 80)):
    h, w = hw
    x = torch.rand((n_img, h, w, c), device=dev)
    out = torch.empty_like(x)
    w9 = torch.rand((9, c), device=dev)
    s = torch.ones(c, device=dev)
    b = torch.zeros(c, device=dev)
    d = L.DwDesc()
    d.inp, d.ldi, d.w9c, d.scale, d.bias = x.data_ptr(), c, w9.data_ptr(), s.data_ptr(), b.data_ptr()
    d.out, d.ldo = out.data_ptr(), c
    d.n_img, d.H, d.W, d.C, d.stride, d.dilation, d.act = n_img, h, w, c, 1, 1, 1
    plan = C.c_void_p(lib.uavsal_plan_create())
    lib.uavsal_plan_add_dw(plan, C.byref(d))
    return plan, (x, out, w9, s, b)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


if __name__ == "__main__":
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    cur = torch.cuda.current_stream()
    for prec, tile, n_img in (("f32", 0, 4), ("f32", 0, 8), ("f16x3", 5, 32), ("f16x3", 6, 32), ("f16x3", 1, 32)):
        gp, keep1 = gemm_plan(n_img, 256, 1536, prec, tile)
        dp, keep2 = dw_plan(n_img, 1536)
        p1, p2 = C.c_void_p(s1.cuda_stream), C.c_void_p(s2.cuda_stream)

        def both():
            s1.wait_stream(cur); s2.wait_stream(cur)
            lib.uavsal_plan_run(gp, 0, -1, p1)
            lib.uavsal_plan_run(dp, 0, -1, p2)
            cur.wait_stream(s1); cur.wait_stream(s2)

        def only(plan, st, ps):
            def f():
                st.wait_stream(cur)
                lib.uavsal_plan_run(plan, 0, -1, ps)
                cur.wait_stream(st)
            return f
        tg, td, tb = timeit(only(gp, s1, p1)), timeit(only(dp, s2, p2)), timeit(both)
        print("%s tile %d, %d frames: GEMM 256->1536 %.1f us, depthwise 1536ch %.1f us, both on two streams %.1f us (sum %.1f)" % (
            prec, tile, n_img, tg, td, tb, tg + td), flush=True)

"""Probe of the LDS-halo depthwise -> projection kernel (fp32): fused launch vs depthwise + projection launches
on the path's 45x80 shapes; also checks the two against each other."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iip_uavsal_saliency_amd import _lib as L, packing as P

lib = L.load()
dev = torch.device("cuda")


def time_plan(adders, iters=20):
    plan = C.c_void_p(lib.uavsal_plan_create())
    for fn, d in adders:
        fn(plan, C.byref(d))
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ms = C.c_float()
    n = len(adders)
    L.check(lib.uavsal_plan_time(plan, 0, n, 3, st, C.byref(ms)), "time")
    L.check(lib.uavsal_plan_time(plan, 0, n, iters, st, C.byref(ms)), "time")
    lib.uavsal_plan_destroy(plan)
    return ms.value * 1e3


def run(n_img, h, w, hid, cout, act=0, use_res=False, stream_k=True, prec="f32"):
    g = torch.Generator().manual_seed(5)
    e = (torch.rand((n_img, h, w, hid), generator=g) * 6).to(dev)
    wd = ((torch.rand((hid, 1, 3, 3), generator=g) - 0.5) * 0.8)
    sd, bd = (torch.rand(hid, generator=g) * 0.5 + 0.75).to(dev), ((torch.rand(hid, generator=g) - 0.5)).to(dev)
    wp = (torch.rand((cout, hid, 1, 1), generator=g) - 0.5) * (2.0 / hid ** 0.5)
    npad = P.roundup(cout, 32)
    s = (torch.rand(npad, generator=g) * 0.5 + 0.75).to(dev)
    b = (torch.rand(npad, generator=g) - 0.5).to(dev)
    res = torch.rand((n_img, h, w, cout), generator=g).to(dev) if use_res else None
    w9 = P.pack_dw_weight(wd).to(dev)
    wpk = P.pack_conv_weight(wp, prec).to(dev)
    wpj = P.pack_conv_weight(wp, "f16x3j").to(dev) if prec == "f16x3" else wpk
    dmid = torch.empty((n_img, h, w, hid), device=dev)
    out_a = torch.zeros((n_img, h, w, cout), device=dev)
    out_b = torch.zeros((n_img, h, w, cout), device=dev)
    ws = torch.zeros(int(lib.uavsal_streamk_workspace_bytes()), dtype=torch.uint8, device=dev)

    dd = L.DwDesc()
    dd.inp, dd.ldi, dd.w9c, dd.scale, dd.bias = e.data_ptr(), hid, w9.data_ptr(), sd.data_ptr(), bd.data_ptr()
    dd.out, dd.ldo = dmid.data_ptr(), hid
    dd.n_img, dd.H, dd.W, dd.C, dd.stride, dd.dilation, dd.act = n_img, h, w, hid, 1, 1, L.ACT_RELU6

    def conv_desc(a, out, fused):
        d = L.ConvDesc()
        d.a, d.lda, d.a_img_stride = a.data_ptr(), hid, h * w
        d.w = wpj.data_ptr() if fused else wpk.data_ptr()
        d.scale, d.bias = s.data_ptr(), b.data_ptr()
        d.out, d.ldc, d.o_img_stride = out.data_ptr(), cout, h * w
        if res is not None:
            d.res, d.ldr, d.r_img_stride = res.data_ptr(), cout, h * w
        d.n_img, d.H, d.W, d.Cin, d.Cout, d.taps = n_img, h, w, hid, cout, 1
        d.prec, d.act, d.epi, d.tile = L.PREC[prec], act, L.EPI_AFFINE, 0
        if fused:
            d.dw_w9c, d.dw_scale, d.dw_bias = w9.data_ptr(), sd.data_ptr(), bd.data_ptr()
            d.dw_stride, d.dw_Hin, d.dw_Win = 1, h, w
        if stream_k and (not fused or os.environ.get("DWPROJ_KSPLIT", "1") != "0"):
            d.sk_ws, d.sk_ws_bytes = ws.data_ptr(), ws.numel()     # stream-K / K-split workspace
        return d
    dp = conv_desc(dmid, out_a, False)
    df = conv_desc(e, out_b, True)
    t_dw = time_plan([(lib.uavsal_plan_add_dw, dd)])
    t_pl = time_plan([(lib.uavsal_plan_add_conv, dp)])
    t_f = time_plan([(lib.uavsal_plan_add_conv, df)])
    torch.cuda.synchronize()
    err = (out_a - out_b).abs().max().item()
    fl = 2.0 * n_img * h * w * hid * cout
    print("%s n=%d %dx%d hid=%d cout=%d act=%d res=%d: dw %.1f + pl %.1f = %.1f us | fused %.1f us (%.1f TFLOP/s, %.0f GB/s of E) | max diff %.2e"
          % (prec, n_img, h, w, hid, cout, act, use_res, t_dw, t_pl, t_dw + t_pl, t_f, fl / t_f / 1e6,
             4.0 * n_img * h * w * hid / t_f / 1e3, err), flush=True)
    return err


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "one":        # one variant, for rocprofv3 --pmc
        run(int(sys.argv[3]) if len(sys.argv) > 3 else 8, 45, 80, 1536, 256, int(sys.argv[2]), False)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "slope":      # per-K-step cost: same launch at three depths
        prec = sys.argv[3] if len(sys.argv) > 3 else "f32"
        for act in ((0, 158, 134, 130, 132, 129, 131) if len(sys.argv) < 3 or sys.argv[2] != "product" else (0,)):
            for hid in (768, 1536, 3072):
                run(8, 45, 80, hid, 256, act, False, prec=prec)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "parts":      # library built with -DUAVSAL_PROBE
        for bits in (0, 2, 4, 6, 14, 30, 1, 3):
            what = ", ".join(n for b, n in ((1, "no MFMAs"), (2, "no depthwise"), (4, "no DMA"), (8, "no fragment loads"),
                                            (16, "no barriers")) if bits & b) or "full"
            print(what)
            run(8, 45, 80, 1536, 256, 128 + bits if bits else 0, False)
        sys.exit(0)
    worst = 0.0
    for sh in [(8, 45, 80, 1536, 256, 0, True), (8, 45, 80, 1920, 256, 0, False), (8, 45, 80, 1536, 1, 2, False),
               (8, 45, 80, 1152, 64, 0, False), (64, 45, 80, 1536, 256, 0, True), (2, 23, 41, 96, 128, 1, False),
               (1, 9, 13, 48, 24, 0, False), (32, 90, 160, 1536, 256, 0, False)]:
        worst = max(worst, run(*sh))
    for sh in [(8, 45, 80, 1536, 256, 0, True), (8, 45, 80, 1536, 1, 2, False), (8, 45, 80, 1152, 64, 0, False),
               (64, 45, 80, 1536, 256, 0, True), (2, 23, 41, 96, 128, 1, False)]:
        worst = max(worst, 0.01 * run(*sh, prec="f16x3"))        # (f16x3 vs f16x3: ~1e-4 apart)
    print("worst diff %.2e" % worst)
    sys.exit(0 if worst < 2e-3 else 1)

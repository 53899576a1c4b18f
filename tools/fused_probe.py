"""uavsal_fused_ir on the six block shapes of MobileNetV2 features[1..7] at the 360x640 map sizes: hipEvent time per launch,
algorithmic GB/s, both patch shapes.   python tools/fused_probe.py [n_img ...]      (UAVSAL_HIP_LIB picks a variant build)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iip_uavsal_saliency_amd import _lib as L, packing as P

lib = L.load()
dev = torch.device("cuda")
BLOCKS = [("features.1", 180, 320, 32, 32, 16, 1, False), ("features.2", 180, 320, 16, 96, 24, 2, True),
          ("features.3", 90, 160, 24, 144, 24, 1, True), ("features.4", 90, 160, 24, 144, 32, 2, True),
          ("features.5", 45, 80, 32, 192, 32, 1, True), ("features.7", 45, 80, 32, 192, 64, 2, True)]


def run(n, h, w, cin, hid, cout, stride, expand, tile, iters=10):
    g = lambda *s: (torch.rand(*s, device=dev) - 0.5)
    x = g(n * h * w, cin)
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    out = torch.empty(n * ho * wo, cout, device=dev)
    keep = [g(cin, hid), g(hid) + 1, g(hid), g(9, hid), g(hid) + 1, g(hid), g(hid, cout), g(cout) + 1, g(cout)]
    d = L.FusedIrDesc()
    d.inp, d.ldi = x.data_ptr(), cin
    if expand:
        d.w1, d.scale1, d.bias1 = keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr()
    d.wd, d.scale_d, d.bias_d = keep[3].data_ptr(), keep[4].data_ptr(), keep[5].data_ptr()
    d.w2, d.scale2, d.bias2 = keep[6].data_ptr(), keep[7].data_ptr(), keep[8].data_ptr()
    if stride == 1 and cin == cout:
        d.res, d.ldr = x.data_ptr(), cin
    d.out, d.ldo = out.data_ptr(), cout
    d.n_img, d.H, d.W, d.Cin, d.hidden, d.Cout, d.stride, d.tile = n, h, w, cin, hid, cout, stride, tile
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    plan = C.c_void_p(lib.uavsal_plan_create())
    L.check(lib.uavsal_plan_add_fused_ir(plan, C.byref(d)), "add")
    ms = C.c_float()
    lib.uavsal_plan_time(plan, 0, 1, 3, st, C.byref(ms))
    L.check(lib.uavsal_plan_time(plan, 0, 1, iters, st, C.byref(ms)), "time")
    lib.uavsal_plan_destroy(plan)
    by = 4.0 * n * (h * w * cin + ho * wo * cout)
    return ms.value * 1e3, by / ms.value / 1e6


if __name__ == "__main__":
    ns = [int(a) for a in sys.argv[1:]] or [8, 64]
    for n in ns:
        tot = {0: 0.0, 1: 0.0, 2: 0.0}
        for b in BLOCKS:
            line = []
            for tile in (0, 1, 2):
                us, gbs = run(n, *b[1:], tile)
                tot[tile] += us * (2 if b[0] == "features.5" else 1)
                line.append("%s %7.1f us %6.0f GB/s" % ({0: "auto", 1: "4x16", 2: "big"}[tile], us, gbs))
            print("n=%d %-10s %s" % (n, b[0], " | ".join(line)), flush=True)
        print("n=%d features.1-7 sum: auto %.1f us, 4x16 %.1f us, big %.1f us" % (n, tot[0], tot[1], tot[2]), flush=True)

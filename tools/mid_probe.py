"""Phase stamps of fused_mid_kernel (workgroup 0, wave 0) from a -DUAVSAL_MID_STAMPS build:
   SRC=fused_mid VARIANTS="stamps:-DUAVSAL_MID_STAMPS" bash tools/build_probe.sh
   UAVSAL_HIP_LIB=tools/_tmp/libuavsal_hip_stamps.so python3 tools/mid_probe.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from iip_uavsal_saliency_amd import _lib as L, ops

lib = L.load()
g = lambda *s: torch.rand(*s) - 0.5
for (n, h, w, cin, hid, cout) in [(8, 23, 40, 64, 384, 64), (8, 23, 40, 96, 576, 96)]:
    x = g(n, h, w, cin).cuda()
    bn = lambda c: (g(c) * 0.5 + 1.0, g(c))
    args = (g(hid, cin, 1, 1), bn(hid), g(hid, 1, 3, 3), bn(hid), g(cout, hid, 1, 1), bn(cout))
    for _ in range(3):
        ops.fused_ir(x, *args, stride=1, residual=cin == cout)
    st = (C.c_ulonglong * 32)()
    assert lib.uavsal_mid_stamps(st) == 0
    v = np.array(list(st), dtype=np.int64)
    t0 = v[0]
    nch = hid // 64
    print("(%d, %d, %d): cycles since kernel start -- x/W1 landed %d, expand(0) done %d" % (cin, hid, cout, v[1] - t0, v[2] - t0))
    prev = v[2]
    for c in range(nch):
        print("   iteration %d: %6d cycles (of which barrier wait %5d)" % (c, v[3 + c] - prev, v[3 + c] - v[16 + c]))
        prev = v[3 + c]
    print("   last projection %d, reduction + store %d, total %d cycles" % (v[13] - prev, v[14] - v[13], v[14] - t0))

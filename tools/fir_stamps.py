"""Phase stamps of fused_ir_kernel (a mid-grid workgroup, wave 0; every stamp behind s_waitcnt 0, so the phases do not overlap each
other inside the wave: costs, not the schedule) from a -DUAVSAL_FIR_STAMPS build:
   SRC=fused_ir VARIANTS="stamps:-DUAVSAL_FIR_STAMPS" bash tools/build_probe.sh
   UAVSAL_HIP_LIB=tools/_tmp/libuavsal_hip_stamps.so python3 tools/fir_stamps.py [n_img]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tools.fused_probe import BLOCKS, run, lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for b in BLOCKS:
    us, gbs = run(n, *b[1:], 0)
    st = (C.c_ulonglong * 128)()
    assert lib.uavsal_fir_stamps(st) == 0
    v = np.array(list(st), dtype=np.int64)
    t0 = v[0]
    print("%s (%.1f us per launch in this build): x loads %d, weights staged %d" % (b[0], us, v[1] - t0, v[2] - v[1]))
    tot = {"wfrag": 0, "expand": 0, "barrier": 0, "dw": 0, "proj": 0}
    nch = 0
    prev = v[2]
    for ch in range(16):
        a = v[8 + 5 * ch: 13 + 5 * ch]
        if a[0] <= prev or a[4] < a[0] or a[0] == 0:
            break
        tot["wfrag"] += a[0] - prev; tot["expand"] += a[1] - a[0]; tot["barrier"] += a[2] - a[1]; tot["dw"] += a[3] - a[2]; tot["proj"] += a[4] - a[3]
        prev = a[4]
        nch += 1
    print("   %d chunks: " % nch + ", ".join("%s %d" % kv for kv in tot.items()) + " cycles; epilogue %d; total %d cycles" % (v[4] - v[3], v[4] - t0))

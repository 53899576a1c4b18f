"""Long-K fp32 / f16x3 GEMM launches for a clock measurement under rocprofv3 --pmc GRBM_GUI_ACTIVE."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_probe import run
for prec in ("f32", "f16x3"):
    for sh in [((45, 80), 8, 4096, 1536, 1), ((45, 80), 8, 256, 1536, 1), ((45, 80), 8, 1536, 256, 1)]:
        ms, tf = run(*sh, prec, 1, iters=30)
        print(prec, sh, "%.1f us %.1f TF" % (ms * 1e3, tf), flush=True)

"""Times uavsal_conv_gemm on chosen shapes (hipEvents on the launch stream) -- tuning aid."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iip_uavsal_saliency_amd import _lib as L, packing as P

lib = L.load()
dev = torch.device("cuda")


def run(M_hw, n_img, cin, cout, taps, prec, tile, iters=20, act=1, scale=True, stream_k=False):
    h, w = M_hw
    a = torch.rand((n_img * h * w, cin), device=dev) * 2 - 1
    wt = (torch.rand((cout, cin, 3 if taps == 9 else 1, 3 if taps == 9 else 1)) - 0.5) * 0.1
    wp = P.pack_conv_weight(wt, prec).to(dev)
    out = torch.empty((n_img * h * w, cout), device=dev)
    s = torch.ones(P.roundup(cout, 32), device=dev)
    b = torch.zeros(P.roundup(cout, 32), device=dev)
    d = L.ConvDesc()
    d.a, d.lda, d.a_img_stride = a.data_ptr(), cin, h * w
    d.w = wp.data_ptr()
    if scale:
        d.scale, d.bias = s.data_ptr(), b.data_ptr()
    d.out, d.ldc, d.o_img_stride = out.data_ptr(), cout, h * w
    d.n_img, d.H, d.W, d.Cin, d.Cout, d.taps = n_img, h, w, cin, cout, taps
    d.prec, d.act, d.epi, d.tile = L.PREC[prec], act, 0, tile
    if stream_k:
        ws = torch.zeros(int(lib.uavsal_streamk_workspace_bytes()), dtype=torch.uint8, device=dev)
        d.sk_ws, d.sk_ws_bytes = ws.data_ptr(), ws.numel()
    plan = C.c_void_p(lib.uavsal_plan_create())
    lib.uavsal_plan_add_conv(plan, C.byref(d))
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ms = C.c_float()
    lib.uavsal_plan_time(plan, 0, 1, 3, st, C.byref(ms))
    L.check(lib.uavsal_plan_time(plan, 0, 1, iters, st, C.byref(ms)), "time")
    lib.uavsal_plan_destroy(plan)
    fl = 2.0 * n_img * h * w * cin * cout * taps
    return ms.value, fl / ms.value / 1e9


if __name__ == "__main__":
    if os.environ.get("PROBE_NOSTORE"):
        for K in (256, 512, 1024, 2048):
            for act in (1, 100):
                ms, tf = run((45, 80), 8, K, 1536, 1, "f32", 1, act=act)
                print("K=%d N=1536 act=%d: %.1f us %.1f TF" % (K, act, ms * 1e3, tf), flush=True)
        for K in (256, 1536):
            for act in (1, 100):
                ms, tf = run((45, 80), 8, K, 256, 1, "f32", 1, act=act)
                print("K=%d N=256 act=%d: %.1f us %.1f TF" % (K, act, ms * 1e3, tf), flush=True)
        sys.exit(0)
    shapes = [((45, 80), 8, 256, 1536, 1), ((45, 80), 8, 1536, 256, 1), ((45, 80), 8, 4096, 1536, 1),
              ((45, 80), 8, 256, 256, 1), ((45, 80), 8, 448, 256, 9), ((45, 80), 1, 256, 256, 9),
              ((180, 320), 8, 16, 96, 1), ((180, 320), 8, 96, 24, 1), ((45, 80), 64, 256, 1536, 1)]
    precs = sys.argv[1].split(",") if len(sys.argv) > 1 else ["f32", "f16x3"]
    for prec in precs:
        for sh in shapes:
            for tile in (1, 2, 4, 5, 6):
                if tile in (5, 6) and (prec == "f32" or sh[3] % 256):
                    continue
                if sh[3] < 64 and tile != 1:
                    continue
                t = 3 if sh[3] <= 32 else tile
                ms, tf = run(*sh, prec, t)
                print("%-6s hw=%s n=%d K=%d N=%d taps=%d tile=%d : %8.1f us  %7.2f TFLOP/s" % (
                    prec, sh[0], sh[1], sh[2], sh[3], sh[4], t, ms * 1e3, tf), flush=True)

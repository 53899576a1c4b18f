"""Per-layer error contribution of single-pass bf16 MFMA inputs, and mixed plans.
Run on the GPU box:  python3 tools/precision_table.py > gpurun_out/r4_precision.txt
Reference for every figure: the REFERENCE's own map for the same inputs (tests/golden/e2e_360x640_T8.npz, produced by
/root/reference/model.py on CPU), 360x640, 1 clip x 8 frames.  `model.prec_overrides` runs single layers in another precision."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iip_uavsal_saliency_amd import UAVSal, synth

g = np.load(os.path.join(ROOT, "tests", "golden", "e2e_360x640_T8.npz"))
T, H, W, seed = 8, 360, 640, int(g["seed"])
h, w = H // 8, W // 8
x = torch.from_numpy(synth.normalize_frames(synth.synth_frames_u8(T, H, W, seed))).cuda()
cb = [torch.from_numpy(synth.gauss_priors(T, h, w)).cuda(), torch.from_numpy(synth.ob_priors(T, h, w, seed=seed)).cuda()]
gold = g["out"]


def run(prec, overrides=None, time_it=False):
    try:
        return _run(prec, overrides, time_it)
    except RuntimeError as e:          # a combination some kernel has no instance for (e.g. the grouped ASPP projection is fp32-only)
        print("    (not runnable: %s)" % str(e)[:90])
        return float("nan"), float("nan")


def _run(prec, overrides=None, time_it=False):
    m = UAVSal(time_dims=T, precision=prec)
    synth.load_synth_weights(m, seed)
    m = m.cuda().eval()
    m.prec_overrides = overrides
    out, _ = m(x, cb, None)
    err = float(np.abs(out.cpu().numpy() - gold).max())
    fps = None
    if time_it:
        xs, cbs = x.view(1, T, 3, H, W), [cb[0].view(1, T, 8, h, w), cb[1].view(1, T, 20, h, w)]
        for _ in range(3):
            m.forward_clips(xs, cbs, None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            m.forward_clips(xs, cbs, None)
        torch.cuda.synchronize()
        fps = T * 20 / (time.perf_counter() - t0)
    return err, fps


GROUPS = [("backbone 8-17 (pw/pl GEMMs)", ["features."]), ("prior nets", ["gauss.", "ob."]), ("ASPP expands + laterals", ["aspp.pw", "aspp1", "conv_lv"]),
          ("conv_last (3x3 448->256)", ["conv_last"]), ("st0", ["st0."]), ("st1", ["st1."]), ("fust", ["fust"]), ("ctx", ["ctx."]),
          ("fucb", ["fucb."]), ("fucbst", ["fucbst"]), ("twa.wx (3x3, hoisted)", ["twa.wx"]), ("twa steps (3x3 recurrence)", ["twa.step"]),
          ("decoder conv_out_st", ["conv_out_st"])]
print("max-abs error of the saliency map vs the reference's own output (360x640, 8 frames); tolerance 1e-3")
for prec in ("f32", "f16x3", "bf16x3", "bf16"):
    e, f = run(prec, None, True)
    print("all layers %-7s: %.3e   %7.1f frames/s" % (prec, e, f))
print("\nONE group in single-pass bf16, everything else exact fp32:")
contrib = {}
for name, keys in GROUPS:
    e, _ = run("f32", {k: "bf16" for k in keys})
    contrib[name] = e
    print("  %-34s %.3e" % (name, e))
print("\nONE group in split f16x3, everything else single-pass bf16 (what the split buys per group):")
for name, keys in GROUPS:
    e, _ = run("bf16", {k: "f16x3" for k in keys})
    print("  %-34s %.3e" % (name, e))
order = sorted(GROUPS, key=lambda nk: contrib[nk[0]])
print("\nmixed plans: f16x3 everywhere, the k least sensitive groups (by the table above) in single-pass bf16:")
ov = {}
for i, (name, keys) in enumerate(order):
    ov.update({k: "bf16" for k in keys})
    e, f = run("f16x3", dict(ov), True)
    print("  + %-32s -> %.3e   %7.1f frames/s   %s" % (name, e, f, "OK" if e <= 1e-3 else "exceeds 1e-3"))
    if e > 2e-2:
        break

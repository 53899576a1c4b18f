"""Soak: the same request N times; every map and state must equal the first bit for bit (no atomics, fixed summation
orders: a race in a kernel's LDS ring or a stream-K / K-split hand-off would show up as a mismatch or a NaN)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iip_uavsal_saliency_amd import UAVSal, synth
from bench import make_clips

dev = torch.device("cuda:0")
T, H, W = 8, 360, 640
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
bad = 0
for prec in ("f32", "f16x3"):
    for clips in (1, 3, 8):
        x, cb = make_clips(clips, T, H, W)
        x = x.to(dev)
        cb = [cb[0].to(dev), cb[1].to(dev)]
        m = UAVSal(time_dims=T, precision=prec)
        synth.load_synth_weights(m, 0)
        m = m.to(dev).eval()
        st = torch.rand((clips, 256, H // 8, W // 8), device=dev)
        ref_o, ref_s = m.forward_clips(x, cb, st)
        torch.cuda.synchronize(dev)
        n = max(10, N // clips)
        mism = 0
        t0 = time.perf_counter()
        for i in range(n):
            o, s = m.forward_clips(x, cb, st)
            if not (torch.equal(o, ref_o) and torch.equal(s, ref_s)):
                mism += 1
        torch.cuda.synchronize(dev)
        m.check_errors()
        finite = bool(torch.isfinite(ref_o).all() and torch.isfinite(ref_s).all())
        print("%s clips=%d: %d runs, %d mismatches, finite=%s, %.1f frames/s incl. the compare" % (
            prec, clips, n, mism, finite, clips * T * n / (time.perf_counter() - t0)), flush=True)
        bad += mism + (0 if finite else 1)
sys.exit(1 if bad else 0)

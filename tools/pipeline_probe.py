"""How much idle capacity does one forward leave?  Independent requests (different videos) issued round-robin on
S host streams, each with its own model replica (own launch plans and buffers), against the same requests one after
another on one stream.  Per-request work is unchanged; only the overlap between consecutive requests differs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iip_uavsal_saliency_amd import UAVSal, synth
from bench import make_clips

dev = torch.device("cuda:0")
T, H, W = 8, 360, 640
for prec in ("f32", "f16x3"):
    for clips in (1, 2):
        x, cb = make_clips(clips, T, H, W)
        x = x.to(dev)
        cb = [cb[0].to(dev), cb[1].to(dev)]
        for S in (1, 2, 3):
            models, streams = [], []
            for i in range(S):
                m = UAVSal(time_dims=T, precision=prec)
                synth.load_synth_weights(m, 0)
                models.append(m.to(dev).eval())
                streams.append(torch.cuda.Stream(dev))
            outs = [None] * S

            def run(n):
                for k in range(n):
                    i = k % S
                    with torch.cuda.stream(streams[i]):
                        outs[i] = models[i].forward_clips(x, cb, None)
            run(2 * S)
            torch.cuda.synchronize(dev)
            n = 30
            t0 = time.perf_counter()
            run(n)
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t0
            for m in models:
                m.check_errors()
            print("%s clips=%d streams=%d: %.1f frames/s (%.3f ms per request)" % (prec, clips, S, clips * T * n / dt, dt / n * 1e3),
                  flush=True)
            del models

# the same through stream.RequestPipeline (replicas share the packed weights)
from iip_uavsal_saliency_amd.stream import RequestPipeline
for prec in ("f32", "f16x3"):
    x, cb = make_clips(1, T, H, W)
    x = x.to(dev)
    cb = [cb[0].to(dev), cb[1].to(dev)]
    m = UAVSal(time_dims=T, precision=prec)
    synth.load_synth_weights(m, 0)
    m = m.to(dev).eval()
    for S in (1, 2):
        pipe = RequestPipeline(m, streams=S)
        for _ in range(4):
            pipe.forward_clips(x, cb, None)
        torch.cuda.synchronize(dev)
        n = 30
        t0 = time.perf_counter()
        for _ in range(n):
            pipe.forward_clips(x, cb, None)
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        pipe.synchronize()
        print("%s RequestPipeline streams=%d: %.1f frames/s" % (prec, S, T * n / dt), flush=True)

"""One-off calibration of the synthetic BatchNorm statistics (test infrastructure).

Runs the oracle once at 96x160, T=4 with forward-pre-hooks that measure the
scalar mean/var of every BatchNorm input (in execution order, so each layer sees
already-calibrated predecessors) and writes them to
`iip_uavsal_saliency_amd/synth_calib.json`.  `synth.synth_tensor` turns these
per-layer scalars into per-channel running_mean/var with a hashed jitter.
Usage:  python oracle/calibrate_synth.py
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from iip_uavsal_saliency_amd import synth  # noqa: E402
from oracle.uavsal_ref import RefUAVSal    # noqa: E402


def main(seed=0, H=96, W=160, T=4):
    torch.set_num_threads(8)
    model = RefUAVSal(time_dims=T)
    synth.load_synth_weights(model, seed, calib={})
    model.eval()
    stats = {}

    def make_hook(name, bn):
        def hook(mod, inp):
            x = inp[0]
            m = float(x.mean())
            v = float(x.var(unbiased=False))
            stats[name] = [m, max(v, 1e-8)]
            mean, var = synth.bn_channel_stats(name, bn.num_features, m, max(v, 1e-8), seed)
            bn.running_mean.copy_(torch.from_numpy(mean))
            bn.running_var.copy_(torch.from_numpy(var))
        return hook

    for name, mod in model.named_modules():
        if isinstance(mod, nn.BatchNorm2d):
            mod.register_forward_pre_hook(make_hook(name, mod))

    h, w = H // 8, W // 8
    x = torch.from_numpy(synth.normalize_frames(synth.synth_frames_u8(T, H, W, seed)))
    cb = [torch.from_numpy(synth.gauss_priors(T, h, w)), torch.from_numpy(synth.ob_priors(T, h, w, seed=seed))]
    taps = {}
    out, _ = model(x, cb, None, taps)
    print("calibrated %d BN layers; map range %.4f..%.4f, logits std %.3f" % (
        len(stats), float(out.min()), float(out.max()), float(taps["logits"].std())))
    path = os.path.join(ROOT, "iip_uavsal_saliency_amd", "synth_calib.json")
    with open(path, "w") as f:
        json.dump({"seed": seed, "size": [H, W, T], "bn": stats}, f, indent=0, sort_keys=True)
    print("wrote", path)


if __name__ == "__main__":
    main()

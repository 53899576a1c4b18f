"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own
`model.py` / `model_convlstm.py` (imported unmodified from /root/reference) on
CPU.  Container-only test tooling: /root/reference does not exist on the GPU
box, only the committed .npz fixtures travel.

`model.py` imports `model_feature`, which needs torchvision (absent) and a
network fetch of ImageNet weights (`mobilenet_v2(pretrained=True)`,
model_feature.py:59).  As SURVEY.md 8(c) prescribes, a stand-in module named
`model_feature` is placed in `sys.modules` whose `ReMobileNetV2` is the
structural restatement of torchvision's MobileNetV2 `.features` (same key
names, 2 223 872 parameters); everything else -- UAVSal, SRF-Net, dwBlock,
STBlock, teConv_sub, ConvTWA(Cell), ConvLSTMCell -- executes from the
reference's source.  Weights and inputs come from
`iip_uavsal_saliency_amd.synth`, so only outputs are stored.

Usage:  PYTHONDONTWRITEBYTECODE=1 python oracle/make_goldens.py
"""
import hashlib
import os
import sys
import types

sys.dont_write_bytecode = True

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")

from iip_uavsal_saliency_amd import synth      # noqa: E402
from oracle.uavsal_ref import _Backbone         # noqa: E402  (torchvision stand-in only)


def import_reference():
    class ReMobileNetV2(nn.Module):
        def __init__(self, name="mobilenet_v2"):
            super().__init__()
            self.features = _Backbone().features

        def forward(self, x):
            x1 = self.features[0:2](x)
            x2 = self.features[2:4](x1)
            x3 = self.features[4:7](x2)
            x4 = self.features[7:14](x3)
            x5 = self.features[14:18](x4)
            return x1, x2, x3, x4, x5

    mf = types.ModuleType("model_feature")
    mf.ReMobileNetV2 = ReMobileNetV2
    mf.ReVGG = ReMobileNetV2
    mf.ReResNet = ReMobileNetV2
    sys.modules["model_feature"] = mf
    sys.path.insert(0, REF)
    import model as ref_model                 # /root/reference/model.py
    import model_convlstm as ref_rnn          # /root/reference/model_convlstm.py
    return ref_model, ref_rnn


def sub(t, stride):
    a = t.detach().contiguous().view(-1).numpy()
    return a[::stride].astype(np.float32).copy()


def inputs(n, H, W, seed=0, t0=0):
    h, w = H // 8, W // 8
    x = torch.from_numpy(synth.normalize_frames(synth.synth_frames_u8(n, H, W, seed, t0)))
    cb = [torch.from_numpy(synth.gauss_priors(n, h, w)), torch.from_numpy(synth.ob_priors(n, h, w, seed=seed))]
    return x, cb, torch.zeros(1, 256, h, w)


def weights_digest(model):
    hsh = hashlib.sha256()
    for k, v in model.state_dict().items():
        hsh.update(k.encode())
        hsh.update(v.detach().contiguous().numpy().tobytes())
    return hsh.hexdigest()


def run_case(ref_model, name, H, W, T, B=1, seed=0, tap_stride=13, state_stride=3, calls=1, cls="UAVSal", bias_type=(1, 1, 1)):
    n = B * T
    lstm = cls == "UAVSAL_LSTM"
    bias_type = list(bias_type)
    model = getattr(ref_model, cls)(cnn_type="mobilenet_v2", time_dims=T, num_stblock=2, bias_type=bias_type,
                             iosize=[H, W, H // 8, W // 8], planes=256, pre_model_path="")
    synth.load_synth_weights(model, seed)
    model.eval()
    taps = {}

    def grab(key):
        def hook(mod, inp, out):
            taps[key] = out[0] if isinstance(out, tuple) else out
        return hook

    model.sfnet.register_forward_hook(grab("sfnet"))
    model.st_layer[0].register_forward_hook(grab("st0"))
    model.st_layer[1].register_forward_hook(grab("st1"))
    tap_keys = ["sfnet", "st0", "st1", "rnn"]
    if any(bias_type):       # (no fusion blocks in a model without priors, model.py:316)
        model.fucb_layer.register_forward_hook(grab("fust_in_cb"))
        model.fucbst_layer.register_forward_hook(grab("prefuse"))
        tap_keys += ["fust_in_cb", "prefuse"]
    model.rnn.register_forward_hook(grab("rnn"))
    model.conv_out_st.register_forward_hook(grab("logits"))

    rec = {"H": H, "W": W, "T": T, "B": B, "seed": seed, "calls": calls, "bias_type": np.array(bias_type),
           "tap_stride": tap_stride, "state_stride": state_stride,
           "weights_sha256": np.frombuffer(bytes.fromhex(weights_digest(model)), dtype=np.uint8)}
    state = None
    with torch.no_grad():
        for c in range(calls):
            x, cb, zero = inputs(n, H, W, seed, t0=c * n)
            if lstm:     # ConvLSTM.forward unpacks hidden_state[0] as (h, c) (model_convlstm.py:204)
                out, st = model(x, cb, [(zero, zero.clone())] if state is None else state)
                rec["cstate" + ("" if c == 0 else f"_call{c}")] = sub(st[1], state_stride)
                state = [(st[0].detach(), st[1].detach())]
            else:
                out, st = model(x, cb, [zero] if state is None else state)
                state = [st[0].detach()]
            sfx = "" if c == 0 else f"_call{c}"
            rec["out" + sfx] = out.numpy().astype(np.float32)
            rec["logits" + sfx] = taps["logits"].numpy().astype(np.float32)
            rec["state" + sfx] = sub(st[0], state_stride)
            rec["state_sum" + sfx] = np.float64(st[0].double().sum().item())
            if c == 0:
                for k in tap_keys:
                    t = taps[k]
                    rec["tap_" + k] = sub(t, tap_stride)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **rec)
    print("%-28s out %.5f..%.5f  logits std %.3f  -> %s (%.0f KB)" % (
        name, rec["out"].min(), rec["out"].max(), rec["logits"].std(), os.path.basename(path),
        os.path.getsize(path) / 1024))


def clip_inputs(C, T, H, W, seed=0, t0=0):
    """C independent clips, clip c seeded with seed + c (the convention of bench.py make_clips and
    tests/test_hip_e2e.py): x [C,T,3,H,W], cb [[C,T,8,h,w],[C,T,20,h,w]]."""
    h, w = H // 8, W // 8
    xs, g, o = [], [], []
    for c in range(C):
        xs.append(torch.from_numpy(synth.normalize_frames(synth.synth_frames_u8(T, H, W, seed + c, t0))))
        g.append(torch.from_numpy(synth.gauss_priors(T, h, w)))
        o.append(torch.from_numpy(synth.ob_priors(T, h, w, seed=seed + c)))
    return torch.stack(xs), [torch.stack(g), torch.stack(o)]


def run_clips_case(ref_model, name, H, W, T, C, seed=0, calls=1, map_stride=1, state_stride=47):
    """Batched-clips semantics of BASELINE configs[2]-[4]: C independent reference
    calls `forward(x_c [T,3,H,W], cb_c, [state_c])` with time_dims=T, each clip carrying its own state
    over `calls` successive calls (Demo_Test.py:75-86).  Stored: the maps (every `map_stride`-th value of
    the flattened [C,T,1,h,w] tensor + the full maps of clip 0), strided state samples, float64 sums."""
    h, w = H // 8, W // 8
    model = ref_model.UAVSal(cnn_type="mobilenet_v2", time_dims=T, num_stblock=2, bias_type=[1, 1, 1],
                             iosize=[H, W, h, w], planes=256, pre_model_path="")
    synth.load_synth_weights(model, seed)
    model.eval()
    rec = {"H": H, "W": W, "T": T, "C": C, "seed": seed, "calls": calls, "map_stride": map_stride,
           "state_stride": state_stride,
           "weights_sha256": np.frombuffer(bytes.fromhex(weights_digest(model)), dtype=np.uint8)}
    states = [torch.zeros(1, 256, h, w) for _ in range(C)]
    with torch.no_grad():
        for call in range(calls):
            x, cb = clip_inputs(C, T, H, W, seed, t0=call * T)
            outs = []
            for c in range(C):
                out, st = model(x[c], [cb[0][c], cb[1][c]], [states[c]])
                states[c] = st[0].detach()
                outs.append(out)
                print("  %s call %d clip %d done" % (name, call, c), flush=True)
            out = torch.stack(outs)                       # [C,T,1,h,w]
            st = torch.cat(states, 0)                     # [C,256,h,w]
            sfx = "" if call == 0 else f"_call{call}"
            rec["out" + sfx] = sub(out, map_stride)
            rec["out_clip0" + sfx] = out[0].numpy().astype(np.float32)
            rec["out_sum" + sfx] = np.float64(out.double().sum().item())
            rec["state" + sfx] = sub(st, state_stride)
            rec["state_sum" + sfx] = np.float64(st.double().sum().item())
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **rec)
    print("%-28s out %.5f..%.5f -> %s (%.0f KB)" % (name, rec["out"].min(), rec["out"].max(),
                                                    os.path.basename(path), os.path.getsize(path) / 1024))


def run_convlstm(ref_rnn, seed=0):
    """One ConvLSTMCell step from the reference's model_convlstm.py (imports as-is)."""
    hid, h, w = 32, 9, 11
    cell = ref_rnn.ConvLSTMCell((h, w), hid, hid, (3, 3), bias=False)
    wgt = synth.synth_tensor("convlstm.rnn_conv.weight", (4 * hid, 2 * hid, 3, 3), seed)
    cell.rnn_conv.weight.data.copy_(torch.from_numpy(wgt))
    mk = lambda nm: torch.from_numpy((synth.hash_normal(nm, hid * h * w, seed)).astype(np.float32)).view(1, hid, h, w)
    x, hp, cp = mk("convlstm.x"), mk("convlstm.h"), mk("convlstm.c")
    with torch.no_grad():
        hn, cn = cell(x, (hp, cp))
    np.savez_compressed(os.path.join(OUT, "convlstm_step.npz"), hid=hid, hw=np.array([h, w]), seed=seed,
                        h_next=hn.numpy(), c_next=cn.numpy())   # x/h/c/weight regenerate from synth
    print("convlstm_step               h_next std %.4f" % hn.std().item())


def main():
    only = sys.argv[1:]
    if not os.path.isdir(REF):
        raise SystemExit("reference not present: goldens can only be generated in the authoring container")
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    ref_model, ref_rnn = import_reference()
    if not only or "clips" in only:
        # BASELINE configs[2]/[3]: 360x640, 8 independent clips x 8 frames (one GPU's share of configs[3])
        run_clips_case(ref_model, "clips_360x640_C8_T8", 360, 640, 8, 8)
        # BASELINE configs[4]: 720x1280, 4 clips x 16 frames, two successive calls with carried state
        run_clips_case(ref_model, "clips_720x1280_C4_T16_two_calls", 720, 1280, 16, 4, calls=2, map_stride=5,
                       state_stride=211)
    if not only or "demo" in only:
        # Demo_Test.py's own call at its real size (Demo_Test.py:110-125: batch_size=4, time_dims=5 -> ONE forward of 20
        # frames at 360x640): the context tiling quirk (model.py:357-361) and the cross-chunk temporal differences
        # (model.py:194-198) at the shape the script runs, two successive calls with the carried state
        run_case(ref_model, "e2e_360x640_B4T5_two_calls", 360, 640, 5, B=4, tap_stride=1999, state_stride=47, calls=2)
    if not only or "bias" in only:
        # constructor values other than the Demo default (model.py:281-324, 346-365): priors dropped one by one, and none
        run_case(ref_model, "e2e_96x160_T4_bias101", 96, 160, 4, bias_type=(1, 0, 1))
        run_case(ref_model, "e2e_96x160_B2T4_bias010", 96, 160, 4, B=2, bias_type=(0, 1, 0), tap_stride=29)
        run_case(ref_model, "e2e_96x160_T4_bias000_two_calls", 96, 160, 4, bias_type=(0, 0, 0), calls=2)
        run_case(ref_model, "e2e_96x160_B2T4_bias001", 96, 160, 4, B=2, bias_type=(0, 0, 1), tap_stride=29)
    if only and "base" not in only:
        return
    run_case(ref_model, "e2e_96x160_T4", 96, 160, 4)
    run_case(ref_model, "e2e_96x160_B4T5", 96, 160, 5, B=4, tap_stride=61)             # Demo_Test.py default chunking
    run_case(ref_model, "e2e_96x160_T4_two_calls", 96, 160, 4, calls=2)  # carried state
    run_case(ref_model, "e2e_72x104_T3", 72, 104, 3)                     # odd sizes: 9x13 -> 5x7 -> 3x4
    run_case(ref_model, "e2e_lstm_96x160_T4_two_calls", 96, 160, 4, calls=2, cls="UAVSAL_LSTM")
    run_case(ref_model, "e2e_288x512_T8", 288, 512, 8, tap_stride=211, state_stride=29)
    run_case(ref_model, "e2e_360x640_T8", 360, 640, 8, tap_stride=331, state_stride=47)
    run_convlstm(ref_rnn)


if __name__ == "__main__":
    main()

"""Multi-GPU equivalence of the clip-sharded path: W ranks, one process per GPU over
RCCL, each running the HIP model on its own clips + ONE all-gather of the maps.  Checked on rank 0:
  * bit for bit against the same shards run one after another on ONE GPU (same per-launch shapes, so the
    same tiles / stream-K partition and the same summation order) -- this is what sharding must preserve;
  * within the fp32 parity tolerance against the whole batch in one `forward_clips` call: the GEMM tile walk
    and stream-K partition depend on the number of images in a launch, so a different batch size changes the
    summation order (measured on one GPU: 8 clips vs the same clips as 2 x 4 differ by up to 1.3e-4 on the
    map of this synthetic network, which amplifies fp32 round-off ~10^3 times).
Needs >= 2 visible GPUs (skipped on the 1-GPU box); the ranks are fresh child processes started before this
process makes any GPU call (device_count() does not initialise HIP here).  The partitioning / gather logic
itself is covered on CPU by the gloo test in test_host_cpu.py."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["UAVSAL_ROOT"])
import numpy as np, torch, torch.distributed as dist
from iip_uavsal_saliency_amd import UAVSal, synth
from iip_uavsal_saliency_amd.parallel import ClipShard, forward_clips_sharded
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
C = int(os.environ["UAVSAL_C"])
T, H, W = (int(v) for v in os.environ.get("UAVSAL_THW", "3,72,104").split(","))
h, w = H // 8, W // 8
share = os.environ.get("UAVSAL_SHARE_GPU") == "1"          # rehearsal on a one-GPU box: every rank on cuda:0, gloo
dev = torch.device("cuda", 0 if share else rank)
torch.cuda.set_device(dev)
if share:
    dist.init_process_group("gloo", rank=rank, world_size=world)
else:
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
assert dist.get_backend() == ("gloo" if share else "nccl") and dist.get_world_size() == world
def clips(first, count):
    xs, g, o = [], [], []
    for c in range(first, first + count):
        xs.append(torch.from_numpy(synth.normalize_frames(synth.synth_frames_u8(T, H, W, c))))
        g.append(torch.from_numpy(synth.gauss_priors(T, h, w)))
        o.append(torch.from_numpy(synth.ob_priors(T, h, w, seed=c)))
    return torch.stack(xs).to(dev), [torch.stack(g).to(dev), torch.stack(o).to(dev)]
model = UAVSal(time_dims=T)
synth.load_synth_weights(model, 0)
model = model.to(dev).eval()
sh = ClipShard(C, world, rank)
x, cb = clips(sh.first, sh.count)                       # only this rank's shard is ever resident
out, st = forward_clips_sharded(model, x, cb, None, total_clips=C)
out2, st2 = forward_clips_sharded(model, x, cb, st, total_clips=C)      # carried local states
torch.cuda.synchronize()
if rank == 0:
    # (a) every rank's shard run alone on this GPU, one after another: bit-identical to the gathered result
    refs, refs2, ok = [], [], True
    for r in range(world):
        xr, cbr = clips(r * sh.count, sh.count)
        o1, s1 = model.forward_clips(xr, cbr, None)
        o2, s2 = model.forward_clips(xr, cbr, s1)
        refs.append(o1); refs2.append(o2)
        if r == 0:
            ok = ok and torch.equal(st, s1) and torch.equal(st2, s2)      # states stay on the owning rank
    ok = ok and torch.equal(out, torch.cat(refs)) and torch.equal(out2, torch.cat(refs2))
    # (b) the whole batch in one call: same values up to summation order
    xa, cba = clips(0, C)
    full, fst = model.forward_clips(xa, cba, None)
    full2, _ = model.forward_clips(xa, cba, fst)
    tol = max(float((out - full).abs().max()), float((out2 - full2).abs().max()))
    ok = ok and tol <= 5e-4
    print("MULTIGPU_RESULT", "OK" if ok else "MISMATCH", tol, flush=True)
dist.barrier()
dist.destroy_process_group()
'''


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_ranks(world, clips, thw, share_gpu=False):
    env = dict(os.environ, UAVSAL_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               WORLD_SIZE=str(world), UAVSAL_C=str(clips), UAVSAL_THW=thw, HSA_ENABLE_IPC_MODE_LEGACY="0",
               UAVSAL_SHARE_GPU="1" if share_gpu else "0")
    procs = []
    for r in range(world):
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)[-4000:]
    assert "MULTIGPU_RESULT OK" in outs[0], outs[0][-2000:]


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_equals_single_gpu_bitwise(world):
    n = torch.cuda.device_count()          # does not initialise the GPU in this process
    if n < world:
        pytest.skip("needs %d GPUs, %d visible" % (world, n))
    if world > 6 and os.environ.get("UAVSAL_SKIP_8_RANKS") == "1":      # opt-OUT: runs wherever 8 GPUs are visible
        pytest.skip("UAVSAL_SKIP_8_RANKS=1")
    _run_ranks(world, 2 * world, "3,72,104")


def test_sharded_benchmark_shape_two_ranks():
    """The benchmarked per-GPU share of BASELINE configs[3] -- 8 clips x 8 frames at 360x640 per rank -- on two
    ranks: gathered maps bit-identical to the two shards run one after another on one GPU, carried local states,
    and within 5e-4 of all 16 clips in one call."""
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip("needs 2 GPUs, %d visible" % n)
    _run_ranks(2, 16, "8,360,640")


def test_two_ranks_sharing_one_gpu_rehearsal():
    """What a one-GPU box CAN execute of the N-rank path: two rank processes, both on cuda:0, rendezvous over gloo, each
    running the HIP model on its own clips through `forward_clips_sharded`, maps gathered through the host.  Same checks
    as the RCCL test (bit-identical to the shards run one after another, carried local states, 5e-4 against the whole batch
    in one call).  The RCCL all-gather itself is the one call this does not reach."""
    if torch.cuda.device_count() < 1:
        pytest.skip("needs a GPU")
    _run_ranks(2, 4, "3,72,104", share_gpu=True)


def test_bench_two_rank_rehearsal_on_one_gpu():
    """`bench.py --gpus 2 --rehearse-on-one-gpu`: the self-launcher, the rank environment, the clip partition, the gather and
    the barrier-bracketed max-over-ranks timing with the HIP engine under them; the line must say it is a rehearsal."""
    import json
    if torch.cuda.device_count() < 1:
        pytest.skip("needs a GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu", "--clips", "1",
                        "--steps", "3", "--warmup", "1", "--windows", "1", "--no-extra", "--no-cpu-baseline", "--no-roofline"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["ranks_seen"] == 2 and r["n_gpus"] == 1 and "rehearsal" in r and r["value"] > 0
    assert r["config"]["clips_per_gpu"] == 1 and r["steps"] == 3


DEVICE_WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["UAVSAL_ROOT"])
import torch
from iip_uavsal_saliency_amd import UAVSal, synth
T, H, W = 3, 72, 104
h, w = H // 8, W // 8
x = torch.from_numpy(synth.normalize_frames(synth.synth_frames_u8(T, H, W)))
cb = [torch.from_numpy(synth.gauss_priors(T, h, w)), torch.from_numpy(synth.ob_priors(T, h, w))]
model = UAVSal(time_dims=T)
synth.load_synth_weights(model, 0)
torch.cuda.set_device(0)                                   # the CURRENT device stays cuda:0 throughout
m1 = model.to("cuda:1").eval()
o1, s1 = m1(x.to("cuda:1"), [c.to("cuda:1") for c in cb], None)      # engine, plan, weights: all on cuda:1
torch.cuda.synchronize("cuda:1")
o1c, s1c = o1.cpu(), s1[0].cpu()
ok = o1.device.index == 1 and s1[0].device.index == 1
m0 = model.to("cuda:0").eval()                             # (moving the module drops the engines and packed weights)
o0, s0 = m0(x.to("cuda:0"), [c.to("cuda:0") for c in cb], None)
torch.cuda.synchronize("cuda:0")
ok = ok and torch.equal(o1c, o0.cpu()) and torch.equal(s1c, s0[0].cpu())
print("DEVICE_RESULT", "OK" if ok else "MISMATCH", flush=True)
'''


def test_engine_on_a_non_current_device():
    """the launch plan's device resources (error word, `done` event, workspaces) and the packed weights
    must live on the device of the frames, not on the current one: a forward on cuda:1 while cuda:0 is current, then
    the same model on cuda:0 (the per-device weight cache must not hand cuda:1 pointers to the cuda:0 plan)."""
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip("needs 2 GPUs, %d visible" % n)
    env = dict(os.environ, UAVSAL_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", DEVICE_WORKER], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    assert "DEVICE_RESULT OK" in p.stdout, p.stdout[-2000:]

"""Multi-GPU equivalence of the clip-sharded path: W ranks, one process per GPU over
RCCL, each running the HIP model on its own clips + ONE all-gather of the maps == the same clips on a
single GPU, bit for bit.  Needs >= 2 visible GPUs (skipped on the 1-GPU box); the ranks are fresh child
processes started before this process makes any GPU call (device_count() does not initialise HIP here).
The partitioning / gather logic itself is covered on CPU by the gloo test in test_host_cpu.py."""
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
C, T, H, W = int(os.environ["UAVSAL_C"]), 3, 72, 104
h, w = H // 8, W // 8
dev = torch.device("cuda", rank)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
assert dist.get_backend() == "nccl" and dist.get_world_size() == world
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
    xa, cba = clips(0, C)                                # single-GPU run of the whole batch, same process
    ref, rst = model.forward_clips(xa, cba, None)
    ref2, rst2 = model.forward_clips(xa, cba, rst)
    ok = (torch.equal(out, ref) and torch.equal(out2, ref2) and torch.equal(st, rst[:sh.count])
          and torch.equal(st2, rst2[:sh.count]))
    print("MULTIGPU_RESULT", "OK" if ok else "MISMATCH", float((out - ref).abs().max()), flush=True)
dist.barrier()
dist.destroy_process_group()
'''


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 4, 8])
def test_sharded_equals_single_gpu_bitwise(world):
    n = torch.cuda.device_count()          # does not initialise the GPU in this process
    if n < world:
        pytest.skip("needs %d GPUs, %d visible" % (world, n))
    if world > 6 and os.environ.get("UAVSAL_ALLOW_8_RANKS") != "1":
        pytest.skip("more than 6 GPU processes at once: only on a whole-node lease (UAVSAL_ALLOW_8_RANKS=1)")
    env = dict(os.environ, UAVSAL_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               WORLD_SIZE=str(world), UAVSAL_C=str(2 * world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = []
    for r in range(world):
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)[-4000:]
    assert "MULTIGPU_RESULT OK" in outs[0], outs[0][-2000:]

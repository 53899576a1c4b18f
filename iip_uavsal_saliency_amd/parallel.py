"""Multi-GPU layer of the UAVSal path: clips are independent units (no cross-clip term in
the reference forward: eval-mode BN, per-call context prior / temporal differences /
recurrence), so a batch of clips is sharded over ranks -- one process per
GPU -- with no collective on the data path, and ONE all-gather of the output maps
(`[C/W, T, 1, h, w]` fp32 per rank, 0.92 MB for 8x8 frames at 45x80) over RCCL/xGMI.
Recurrent states stay on the owning rank (a video is pinned to a rank)."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch
import torch.distributed as dist


@dataclass(frozen=True)
class ClipShard:
    """Contiguous block partition: rank r owns clips [first, first + count)."""
    total_clips: int
    world_size: int
    rank: int

    def __post_init__(self):
        if self.world_size < 1 or not (0 <= self.rank < self.world_size):
            raise ValueError("bad rank/world_size")
        if self.total_clips % self.world_size:
            raise ValueError("total_clips (%d) must be a multiple of world_size (%d): the all-gather "
                             "exchanges equal-sized blocks" % (self.total_clips, self.world_size))

    @property
    def count(self) -> int:
        return self.total_clips // self.world_size

    @property
    def first(self) -> int:
        return self.rank * self.count

    def local(self, t: torch.Tensor) -> torch.Tensor:
        """This rank's block of a `[total_clips, ...]` tensor."""
        return t[self.first:self.first + self.count]


def gather_maps(local_out: torch.Tensor, gathered: Optional[torch.Tensor] = None) -> torch.Tensor:
    """All-gather the per-rank output maps `[c, T, 1, h, w]` into `[c * world, T, 1, h, w]` (rank
    order == clip order).  One collective; RCCL (`nccl` backend) on GPUs, gloo on CPU tests."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        if gathered is None:
            return local_out
        gathered.copy_(local_out)
        return gathered
    world = dist.get_world_size()
    local_out = local_out.contiguous()
    if gathered is None:
        gathered = torch.empty((local_out.shape[0] * world,) + tuple(local_out.shape[1:]),
                               dtype=local_out.dtype, device=local_out.device)
    if dist.get_backend() == "gloo":
        if local_out.is_cuda:       # gloo has no device all-gather: through the host (one-GPU rehearsals of the rank protocol only)
            parts = [torch.empty(local_out.shape, dtype=local_out.dtype) for _ in range(world)]
            dist.all_gather(parts, local_out.cpu())
            gathered.copy_(torch.cat(parts))
        else:
            dist.all_gather(list(gathered.chunk(world, 0)), local_out)
    else:
        dist.all_gather_into_tensor(gathered, local_out)
    return gathered


def forward_clips_sharded(model, x, cb, states_local=None, total_clips: Optional[int] = None):
    """Data-parallel `forward_clips`: this rank computes its own clips and the maps of all ranks are
    exchanged with one all-gather.

    `x` / `cb` are either this rank's LOCAL shard (`x [C/W,T,3,H,W]`; pass `total_clips=C`) -- the
    production form: a rank only ever holds 1/W of the frames (configs[3]: 8 of 64 clips, 177 MB of
    1.4 GB) -- or, with `total_clips=None`, the full batch `[C,T,...]` from which the rank's contiguous
    block is sliced (convenient for small tests).  `states_local` `[C/W,256,h,w]` never leaves the rank.
    Returns (all maps `[C,T,1,h,w]` gathered on every rank in clip order, this rank's states)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if total_clips is None:
        sh = ClipShard(x.shape[0], world, rank)
        x, cb = sh.local(x), [sh.local(cb[0]), sh.local(cb[1])]
    else:
        sh = ClipShard(total_clips, world, rank)
        if x.shape[0] != sh.count or cb[0].shape[0] != sh.count or cb[1].shape[0] != sh.count:
            raise RuntimeError("rank %d owns %d of %d clips, got a shard of %d" % (rank, sh.count, total_clips, x.shape[0]))
    out, st = model.forward_clips(x, cb, states_local)
    return gather_maps(out), st

"""Host side of the HIP path: turns a `UAVSal` parameter tree into a native launch plan.

Built once per (device, clip count, sequence length, frame size, precision):
  1. folds every BatchNorm and packs every conv weight (packing.py), uploads them;
  2. lays the NHWC fp32 activation buffers out in HBM (concatenations become channel
     slices of one wider buffer; the two 6x-expanded hidden tensors of an inverted
     residual block live in two scratch buffers shared by all blocks);
  3. records every kernel launch of reference `UAVSal.forward` (model.py:341-375) into a
     `uavsal_plan` (C ABI, include/uavsal_hip.h) that is then replayed natively --
     as a launch loop or as one captured hipGraph.
`run()` only stages the caller's tensors and launches the plan on torch's current
stream.  PyTorch is used for device memory and streams, not for arithmetic.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional

import torch

from . import _lib as L
from . import packing as P
from . import synth


def _down(n: int) -> int:
    return (n - 1) // 2 + 1      # 3x3, stride 2, pad 1


class V:
    """A channel slice [coff, coff+C) of an NHWC buffer `[n*h*w, ld]`.  `sp`: the buffer's split shadow (fp16,
    `[pixel][ld/32][hi 32 | lo 32]`, include/uavsal_hip.h) when some GEMM stages this tensor pre-split; `t` is
    None for a tensor that only exists as its shadow (depthwise outputs)."""
    __slots__ = ("t", "ld", "coff", "n", "h", "w", "c", "sp", "key")

    def __init__(self, t, n, h, w, c, ld=None, coff=0, sp=None, key=None):
        self.t, self.n, self.h, self.w, self.c = t, n, h, w, c
        self.ld = ld if ld is not None else c
        self.coff = coff
        self.sp, self.key = sp, key

    @property
    def ptr(self):
        return None if self.t is None else self.t.data_ptr() + 4 * self.coff

    @property
    def sp_ptr(self):
        """Address of this view inside the shadow: `coff` = pixel offset * ld + channel offset (a multiple of 32)."""
        pix, ch = divmod(self.coff, self.ld)
        assert ch % 32 == 0 and self.ld % 32 == 0
        return self.sp.data_ptr() + 2 * (pix * 2 * self.ld + (ch // 32) * 64)

    def slice(self, coff, c):
        return V(self.t, self.n, self.h, self.w, c, self.ld, self.coff + coff, self.sp, self.key)

    def frames(self, first, count):
        """Images [first, first+count) as a view (pointer offset only)."""
        return V(self.t, count, self.h, self.w, self.c, self.ld, self.coff + first * self.h * self.w * self.ld,
                 self.sp, self.key)


# expanded values (pixels x hidden channels) from which the fused depthwise -> projection launch beats depthwise +
# projection launches (tools/dwproj_probe.py: 2 x 23 x 41 x 96 loses).  Round 2 had 8 x 45 x 80 x 512 here, which kept the
# 384-hidden blocks at 45x80 (temporal sub-blocks, prior nets) unfused at one clip: fused they take 29-32 us instead of 44-46
# (one clip fp32 4.521 -> 4.452 ms, f16x3 3.25 -> 3.19)
FUSE_DW_MIN_WORK = int(os.environ.get("UAVSAL_FUSE_DW_MIN_WORK", str(1 << 20)))
# ... and the share of a map's 8 x 16 pixel patches that lies outside the map must be small: the kernel computes whole
# patches (45x80: 1.07, 23x40: 1.25, 12x20: 2.13).  Eight clips, fp32, features.8-17 on the 23x40 / 12x20 maps: 1259 us fused
# against 911 us as depthwise + projection launches (features.17 alone 282 vs 128)
FUSE_DW_MAX_WASTE = float(os.environ.get("UAVSAL_FUSE_DW_MAX_WASTE", "1.15"))


# the mid-channel fused block kernel (csrc/fused_mid.hip): workgroups (4 x 8 output patches) of a launch for which it is taken
MID_MIN_WGS = int(os.environ.get("UAVSAL_MID_MIN_WGS", "1"))
MID_MAX_WGS = int(os.environ.get("UAVSAL_MID_MAX_WGS", "288"))


# features[14..17] (12x20 maps at 360x640: 12 launches of 6-22 us, none of them a round of the chip) as two half-batch chains on two lanes
F16X3_WINO_STEPS = int(os.environ.get("UAVSAL_F16X3_WINO_STEPS", "1"))
WINO_TAIL_PLANES = int(os.environ.get("UAVSAL_WINO_TAIL_PLANES", "0"))
WINO_SEG = int(os.environ.get("UAVSAL_WINO_SEG", "0"))         # see the SRF-Net head in Engine._build (measured: not faster, off)
TAIL_SPLIT = os.environ.get("UAVSAL_TAIL_SPLIT", "0") == "1"
TAIL_SPLIT_FROM = int(os.environ.get("UAVSAL_TAIL_SPLIT_FROM", "14"))
TAIL_SPLIT_MAX_FRAMES = int(os.environ.get("UAVSAL_TAIL_SPLIT_MAX_FRAMES", "16"))
PRIORS_OB_LANE = int(os.environ.get("UAVSAL_PRIORS_OB_LANE", "1"))      # 1: both prior nets on lane 1 (two event operations fewer on the main stream: 4.29 -> 4.26 ms at one clip); 2: a lane each
ASPP_DW_MERGE = os.environ.get("UAVSAL_ASPP_DW_MERGE", "1") == "1"      # 0: the three dilated ASPP depthwise convs as three launches on three lanes
DW_DOT = os.environ.get("UAVSAL_DW_DOT", "1") == "1"       # 0: the one-channel projection of conv_out_st as a dwproj GEMM + reduce launch
_TILE_OVERRIDE = {kv.split("=")[0]: int(kv.split("=")[1]) for kv in os.environ.get("UAVSAL_TILE_OVERRIDE", "").split(",") if "=" in kv}


def _dwproj_patch_waste(h, w):
    return ((h + 7) // 8 * 8) * ((w + 15) // 16 * 16) / float(h * w)


BLOCK_CHUNK_BYTES = int(float(os.environ.get("UAVSAL_BLOCK_CHUNK_GB", "6")) * (1 << 30))     # see Engine.ir_block
ARENA = os.environ.get("UAVSAL_ARENA", "1") == "1"          # 0: one allocation per activation for the life of the plan (rounds 1-4)
ARENA_ALIGN = 1024                                            # floats (4 KB): every arena buffer starts on a page


class _ArenaRef:
    """An activation's place in the engine's arena: `numel` floats at `off`, live over the recorded ops [first, last]
    (positions on the main lane's timeline; a use on a side lane counts from that lane's fork to its join).  Stands where a
    tensor stood in `V.t`, so every view of the buffer shares it; `data_ptr()` refuses to hand out an address outside the
    live range while a plan is being recorded -- a recorder that forgot to declare a use fails there, at build time."""
    __slots__ = ("eng", "aid", "numel_", "off", "first", "last", "pinned", "lkey", "lfirst", "llast")

    def __init__(self, eng, aid, numel):
        self.eng, self.aid, self.numel_ = eng, aid, int(numel)
        self.off, self.first, self.last, self.pinned = None, None, None, False
        # lkey: (lane, index of its fork) while every use so far was recorded on that lane between that fork and its join
        # (its launches are then ordered among themselves on one stream: [lfirst, llast] in recording order), else "mixed"
        self.lkey, self.lfirst, self.llast = None, None, None

    def numel(self):
        return self.numel_

    def data_ptr(self):
        e = self.eng
        if e._dry:
            return 0
        lo, hi = (self.lfirst, self.llast) if isinstance(self.lkey, tuple) else (self.first, self.last)
        if e._recording and not self.pinned and not (lo <= e._lop <= hi):
            raise RuntimeError("arena: %r is addressed by op %d outside its live range [%d, %d] -- a recorder did not declare "
                               "this use (Engine._touch)" % (self.aid, e._lop, lo, hi))
        return e._arena.data_ptr() + 4 * self.off

    def tensor(self):
        return self.eng._arena[self.off:self.off + self.numel_]


def arena_conflict(a, b):
    """Are two buffers `(numel, first, last[, lane key, lane first, lane last])` ever live together?  [first, last] are positions
    on the main lane's timeline (a side-lane use counts from the fork to the join); two buffers used ONLY on the same side lane
    between the same fork and join are ordered by that lane's stream, so for them the recording-order ranges decide."""
    if a[2] < b[1] or b[2] < a[1]:
        return False
    if len(a) > 3 and len(b) > 3 and a[3] is not None and a[3] == b[3] and isinstance(a[3], tuple):
        return not (a[5] < b[4] or b[5] < a[4])
    return True


def plan_arena(bufs, align=ARENA_ALIGN):
    """Offsets for buffers `[(numel, first, last[, lane key, lane first, lane last]), ...]` such that two buffers that are ever
    live together (`arena_conflict`) never overlap: biggest first, each at the lowest aligned offset free of every already-placed
    buffer it conflicts with.  Returns (offsets, total floats, lower bound = the largest sum of sizes live at one main-lane position)."""
    order = sorted(range(len(bufs)), key=lambda i: (-bufs[i][0], bufs[i][1]))
    placed, offs = [], [0] * len(bufs)
    rnd = lambda n: (n + align - 1) // align * align
    for i in order:
        n, f, l = bufs[i][:3]
        busy = sorted((o, o + rnd(bufs[j][0])) for (o, j) in placed if arena_conflict(bufs[i], bufs[j]))
        at = 0
        for lo, hi in busy:
            if at + rnd(n) <= lo:
                break
            at = max(at, hi)
        offs[i] = at
        placed.append((at, i))
    total = max([o + rnd(bufs[j][0]) for (o, j) in placed], default=0)
    # lower bound: at a main-lane position, everything live there -- of the buffers that are private to one side lane only the
    # largest set that is live together in that lane's own order
    events = sorted(set(b[1] for b in bufs))
    bound = 0
    for t in events:
        live = [b for b in bufs if b[1] <= t <= b[2]]
        tot = sum(rnd(b[0]) for b in live if not (len(b) > 3 and isinstance(b[3], tuple)))
        lanes = {}
        for b in live:
            if len(b) > 3 and isinstance(b[3], tuple):
                lanes.setdefault(b[3], []).append(b)
        for grp in lanes.values():
            tot += max(sum(rnd(c[0]) for c in grp if c[4] <= u <= c[5]) for u in set(c[4] for c in grp))
        bound = max(bound, tot)
    return offs, total, bound


class Engine:
    def __init__(self, model, device, n_seq, seq_len, H, W, ctx_T, ctx_mode="tile",
                 precision="f32", taps=False, in_dtype=torch.float32, use_graph=False, fuse_dw=None,
                 use_lanes=True, stream_k=True, sync_errors=True, persistent=False,
                 wcache=None, static_priors=False, plan_only=False):
        """`plan_only`: sizing pass and arena placement only -- no device, nothing allocated, nothing recorded (memory planning
        questions and the CPU tests of the arena: `arena_stats`, `arena_layout()`)."""
        if precision not in L.PREC:
            raise ValueError("precision must be one of %s" % list(L.PREC))
        self.lib = L.load()
        self.device = torch.device(device)
        self.plan_only = bool(plan_only)
        if self.device.type != "cuda" and not self.plan_only:
            raise RuntimeError("Engine needs a cuda (ROCm) device; there is no CPU fallback")
        self.model, self.n_seq, self.seq_len = model, n_seq, seq_len
        self.N = n_seq * seq_len
        self.H, self.W = H, W
        self.ctx_T, self.ctx_mode = ctx_T, ctx_mode
        self.prec_name, self.prec = precision, L.PREC[precision]
        self.keep_taps = taps
        self.in_dtype = in_dtype
        # the caller's priors are ONE map set for every frame (what the reference's own caller builds: np.repeat of the prior
        # file over b_s frames, utils_data.py:466-467, 601-602; recognised by the model from a zero frame stride, never from the
        # values): the two prior nets run on one frame and their output is broadcast over the frames
        self.static_priors = bool(static_priors)
        self.use_graph = use_graph
        self.stream_k = bool(stream_k)
        # device-side errors (a stream-K hand-off that timed out) are never silent: the guard op at the end of
        # the plan overwrites the outputs with NaN, and `run` raises -- before it returns when `sync_errors`
        # (one event wait per call), else at the next call / `check()` once the run is known to be over
        self.sync_errors = bool(sync_errors)
        self._sk_ws = {}
        self._err = None
        self._first_run_verified = False
        self._sk_debug = tuple(getattr(model, "_sk_debug", (0, 0)))
        # fused depthwise->projection GEMM (uavsal_conv_desc.dw_*): D never reaches HBM.
        #   None (default): the LDS-halo kernel (dwproj_kernel, fp32 and f16x3) on the blocks where it wins -- stride 1,
        #     dilation 1, hidden % 16 == 0 and at least FUSE_DW_MIN_WORK expanded values (the 45x80 / 90x160 dwBlocks
        #     of the head and the decoder: profiles/r2_dwproj.md); other precisions keep the three-launch form;
        #   True: every dilation-1 block with an expand conv (strides 2 and the 16-bit precisions then run the
        #     round-1 register-staged loader, which is correct but ~1.5x slower: tests only);  False: never.
        self.fuse_dw = fuse_dw if fuse_dw is None else bool(fuse_dw)
        if self.N % ctx_T:
            raise RuntimeError("frame count %d is not a multiple of time_dims %d" % (self.N, ctx_T))
        self.h = _down(_down(_down(H)))
        self.w = _down(_down(_down(W)))
        self._keep: List[torch.Tensor] = []        # buffers kept alive
        # packed device weights, keyed by (kind, id(module), ...): one copy per model, shared by its engines
        self._wcache: Dict[tuple, object] = wcache if wcache is not None else {}
        # persistent-state mode: the recurrent state lives in `hprev` (NHWC) across calls, see run()
        self.persistent = bool(persistent)
        # f16x3: tensors that an eligible GEMM consumes are ALSO kept as split shadows (hi/lo fp16 planes) written
        # by their producers, so that GEMM stages both operands by LDS-DMA with no conversion work.  The sizing
        # pass finds out which buffers are wanted (`_split_want`) and which cannot have one because a producer
        # does not write shadows (`_no_shadow`).
        presplit = getattr(model, "presplit", None)
        # per-layer precision (diagnostics: profiles/r4_precision.md): {op-name prefix: precision}, longest prefix wins; the
        # GEMM of that op (and the weights packed for it) then run in that precision, everything else in the plan's
        self.prec_overrides = dict(getattr(model, "prec_overrides", None) or {})
        for v in self.prec_overrides.values():
            if v not in L.PREC:
                raise ValueError("prec_overrides: unknown precision %r" % (v,))
        self.split_mode = (precision == "f16x3" and not self.prec_overrides
                           and (bool(presplit) if presplit is not None else n_seq >= 4))
        # exact-fp32 mode: the dense 3x3 convs (conv_last, the ConvTWA gate conv) as Winograd F(2x2, 3x3)
        # (model.winograd, default on; UAVSAL_WINOGRAD=0 switches it off, UAVSAL_WINOGRAD_STEPS = 0 / 8 / 11: the per-step
        # convolutions of the recurrence too, with that GEMM tile)
        self.winograd = ((precision == "f32" or bool(getattr(model, "prec_overrides", None))) and bool(getattr(model, "winograd", True))
                         and os.environ.get("UAVSAL_WINOGRAD", "1") != "0")
        self.winograd_steps = int(os.environ.get("UAVSAL_WINOGRAD_STEPS", "-1"))     # -1: by the number of clips (below)
        # output tile of the transforms: 2 = F(2x2, 3x3), 4 = F(4x4, 3x3); for the all-frames convs / for the recurrence steps
        # NOTE: with the defaults the arithmetic of "exact fp32" depends on the number of clips in the call -- F(2x2) steps below
        # four clips, F(4x4) (coefficients up to 8, ~20x less accurate per conv, map moves by ~5e-5) from four up -- so the same
        # clip gives maps that differ by ~1e-4 when batched differently (all inside the 5e-4 gate).  `model.winograd_r = 2`
        # (also for the steps) is the strict setting: F(2x2) everywhere, whatever the batch; `model.winograd = False`: direct.
        self.winograd_r = int(getattr(model, "winograd_r", None) or os.environ.get("UAVSAL_WINOGRAD_R", "4"))
        self.winograd_step_r = int(getattr(model, "winograd_step_r", None) or (2 if getattr(model, "winograd_r", None) == 2 else 0)
                                   or os.environ.get("UAVSAL_WINOGRAD_STEP_R", "0"))   # 0: by the number of clips
        self.fuse_blocks = bool(getattr(model, "fuse_blocks", True))
        self._split_want = set()
        self._no_shadow = set()
        self.ops_meta: List[dict] = []
        self._op_idx: Dict[str, int] = {}
        # launch-loop mode reads the caller's tensors in place and writes straight into fresh outputs
        # (uavsal_plan_patch_ptr); a captured graph replays fixed addresses and keeps the staging copies
        self.inplace = not use_graph
        self._hold = None
        self.stage_ranges: Dict[str, tuple] = {}
        self.named: Dict[str, V] = {}
        self._scratch_need: Dict[tuple, int] = {}
        self._scratch: Dict[tuple, torch.Tensor] = {}
        self._lane = 0
        self.plan = None
        # activation arena (liveness-based): see _buf / _touch / _place_arena
        self.use_arena = ARENA and bool(getattr(model, "arena", True))
        self.arena_debug = bool(getattr(model, "arena_debug", False)) or os.environ.get("UAVSAL_ARENA_DEBUG", "0") == "1"
        self._refs: Dict[object, _ArenaRef] = {}
        self._arena = None
        self._recording = False
        self._lop = -1                        # logical op index (poison fills of the debug mode do not count)
        self._lane_open: Dict[int, int] = {}
        self._lane_refs: Dict[int, set] = {}
        self._scr_serial = 0
        self.arena_stats: Dict[str, float] = {}
        # everything below allocates on, or creates native objects for, the CURRENT device (the plan's error word, its
        # `done` event, workspaces, occupancy queries): make that the engine's device, whatever the caller's is
        if self.plan_only:
            self._dry = self._recording = True
            self._build()
            self._close_lanes()
            if self.use_arena:
                self._place_arena()
            return
        with torch.cuda.device(self.device):
            self._init_on_device(use_lanes)

    def _init_on_device(self, use_lanes, resume=False):
        # pass 1 sizes the shared scratch and the arena, pass 2 records the launches (`resume`: pass 1 already ran -- plan_only)
        if not resume:
            self._dry = True
            self._recording = True
            self._build()
            self._close_lanes()
        self._split_want -= self._no_shadow
        if self.use_arena and (not resume or self._arena is None):
            self._place_arena()
        for k, need in self._scratch_need.items():
            # (the Winograd V planes are zero-filled once: their padding rows are multiplied by the GEMM, never read back)
            alloc = torch.zeros if k[0] == "WV" else torch.empty
            self._scratch[k] = alloc(max(need, 4), dtype=torch.float16 if k[0] == "Ds" else torch.float32,
                                     device=self.device)
        self._lane = 0
        self._dry = False
        self._lop, self._scr_serial, self._lane_open, self._lane_refs = -1, 0, {}, {}
        self._poison_done = set()
        self.ops_meta, self.stage_ranges, self.named, self._op_idx = [], {}, {}, {}
        self.plan = C.c_void_p(self.lib.uavsal_plan_create())
        if not self.plan:
            raise RuntimeError("uavsal_plan_create failed")
        self._err = self.lib.uavsal_plan_error_word(self.plan)
        self._build()
        self._flush_poison(final=True)
        self._recording = False
        self.use_lanes = bool(use_lanes)
        L.check(self.lib.uavsal_plan_enable_lanes(self.plan, 1 if self.use_lanes else 0), "plan_enable_lanes")
        self._graph_ready = False

    def __del__(self):
        try:
            if getattr(self, "plan", None) and not getattr(self, "plan_only", False):
                self.lib.uavsal_plan_destroy(self.plan)
                self.plan = None
        except Exception:
            pass

    # ------------------------------------------------------------------ memory helpers
    def _buf(self, name, n, h, w, c, pinned=False) -> V:
        """A named NHWC activation.  With the arena (default) it is `n*h*w*c` floats of ONE pool, placed so that it shares
        addresses only with buffers it is never live together with (first declared use .. last declared use of the recorded
        plan); `pinned`: survives the call (the resident recurrent state), its own allocation."""
        sp = None
        numel = n * h * w * c
        if self.use_arena and not pinned:
            t = self._ref(name, numel)
        elif self._dry:
            t = _Fake()
        else:
            t = torch.empty(numel, dtype=torch.float32, device=self.device)
            self._keep.append(t)
        if not self._dry and name in self._split_want and c % 32 == 0:
            # NaN-filled, not empty: if a producer that cannot write shadows were ever added without entering
            # its output in `_no_shadow`, the GEMM reading this shadow would multiply NaNs -- the first run of the
            # plan then fails loudly (run(): `_verify_first_run`) instead of returning plausible wrong maps.
            # (Shadows stay outside the arena for that reason: a recycled range would hold somebody's finite data)
            sp = torch.full((2 * numel,), float("nan"), dtype=torch.float16, device=self.device)
            self._keep.append(sp)
        v = V(t, n, h, w, c, sp=sp, key=name)
        if name:
            self.named[name] = v
        return v

    # ---- activation arena -------------------------------------------------------------------------------------------
    def _ref(self, aid, numel) -> _ArenaRef:
        if self._dry:
            if aid in self._refs:
                raise RuntimeError("arena: buffer %r declared twice" % (aid,))
            r = self._refs[aid] = _ArenaRef(self, aid, numel)
            return r
        r = self._refs.get(aid)
        if r is None or r.numel_ != int(numel):
            raise RuntimeError("arena: buffer %r of the recording pass was not (or differently) declared in the sizing pass" % (aid,))
        return r

    def _touch(self, *vs):
        """Declare that the op being recorded (the last `_meta`) reads or writes these views.  Sizing pass: grows the live
        range of their arena buffers -- on a side lane from the lane's fork (it may start right there) to, at its join, the
        join (it may still be running until then)."""
        if not self._dry:
            return
        for v in vs:
            r = getattr(v, "t", None) if v is not None else None
            if not isinstance(r, _ArenaRef):
                continue
            lo = hi = self._lop
            key = "main"
            if self._lane != 0:
                lo = self._lane_open.get(self._lane, lo)
                self._lane_refs.setdefault(self._lane, set()).add(r)
                key = (self._lane, self._lane_open.get(self._lane, -1))
            r.first = lo if r.first is None else min(r.first, lo)
            r.last = hi if r.last is None else max(r.last, hi)
            r.lkey = key if r.lkey in (None, key) else "mixed"
            r.lfirst = self._lop if r.lfirst is None else min(r.lfirst, self._lop)
            r.llast = self._lop if r.llast is None else max(r.llast, self._lop)

    def _close_lanes(self):
        for lane in list(self._lane_refs):          # (a lane the plan never joined: live to the end)
            for r in self._lane_refs.pop(lane):
                r.last = max(r.last, len(self.ops_meta))
        self._lane_open = {}

    TAP_NAMES = ("c3", "c4", "c5", "sfnet", "st0", "st1", "fust_in_cb", "prefuse", "rnn")

    def _place_arena(self):
        refs = list(self._refs.values())
        last_op = len(self.ops_meta)
        for r in refs:
            if r.first is None:                      # declared, never used by an op: keep it addressable for the whole plan
                r.first, r.last, r.lkey, r.lfirst, r.llast = 0, last_op, "mixed", 0, last_op
        if self.keep_taps:                           # read back after the run (Engine.tap)
            for k in self.TAP_NAMES:
                v = self.named.get(k)
                if v is not None and isinstance(v.t, _ArenaRef):
                    v.t.last, v.t.lkey = last_op, "mixed"
        offs, total, bound = plan_arena([(r.numel_, r.first, r.last, r.lkey, r.lfirst, r.llast) for r in refs])
        for r, o in zip(refs, offs):
            r.off = o
        self.arena_stats = {"arena_mb": total * 4 / 1e6, "live_bound_mb": bound * 4 / 1e6,
                            "unshared_mb": sum(r.numel_ for r in refs) * 4 / 1e6, "buffers": len(refs)}
        if self.plan_only:
            return
        self._arena = torch.empty(max(total, 4), dtype=torch.float32, device=self.device)
        if self.arena_debug:
            self._arena.fill_(float("nan"))

    def arena_layout(self):
        """[(buffer id, offset, floats, first op, last op, lane key, first / last op in recording order)] of the arena, by offset
        (engine.arena_conflict takes `t[2:]`)."""
        return sorted(((r.aid, r.off, r.numel_, r.first, r.last, r.lkey, r.lfirst, r.llast) for r in self._refs.values()),
                      key=lambda t: (t[1], t[3]))

    def _flush_poison(self, final=False):
        """Debug mode: once the op that ends a buffer's live range has been recorded -- and before anything of the next op,
        a fork included -- the range is filled with NaN on the main lane, so a use after release cannot go unnoticed."""
        if not (self.arena_debug and self.use_arena) or self._dry:
            return
        def end_of(r):          # last logical op that may touch the buffer
            return r.llast if isinstance(r.lkey, tuple) else r.last
        due = [r for r in self._refs.values() if r not in self._poison_done and not r.pinned
               and (final or end_of(r) < self._lop + 1)]
        if not due:
            return
        cur = self._lane
        for r in sorted(due, key=lambda r_: r_.off):
            self._poison_done.add(r)
            if final and end_of(r) >= self._lop:     # still live at the end of the plan (taps, the history the state is read from)
                continue
            # a buffer private to a side lane is released in that lane's own order: its fill goes on that lane (while the lane is
            # open: behind its last launch there, in front of whatever the lane runs next), everything else on the main lane
            lane = r.lkey[0] if isinstance(r.lkey, tuple) and self._lane_open.get(r.lkey[0]) == r.lkey[1] else 0
            if lane != cur:
                L.check(self.lib.uavsal_plan_set_lane(self.plan, lane), "plan_set_lane")
                cur = lane
            self._op_idx["poison:%s" % (r.aid,)] = len(self.ops_meta)
            self.ops_meta.append(dict(kind="poison", name="poison:%s" % (r.aid,), flops=0.0, bytes=4.0 * r.numel_, lane=lane))
            d = L.FillDesc()
            d.out, d.n, d.bits = self._arena.data_ptr() + 4 * r.off, r.numel_, 0x7FC00000
            self._add(self.lib.uavsal_plan_add_fill, d, "plan_add_fill")
        if cur != self._lane:
            L.check(self.lib.uavsal_plan_set_lane(self.plan, self._lane), "plan_set_lane")

    def _scr_split(self, n, h, w, c) -> V:
        """A depthwise output that exists only as its split shadow (scratch, per lane)."""
        numel = 2 * n * h * w * c
        k = ("Ds", self._lane)
        if self._dry:
            self._scratch_need[k] = max(self._scratch_need.get(k, 0), numel)
            return V(None, n, h, w, c, sp=_Fake(), key=k)
        return V(None, n, h, w, c, sp=self._scratch[k], key=k)

    def _would_split(self, n_img, h, w, cin, cout, taps, act, has_res, ldc=None, ldr=None) -> bool:
        """Would `uavsal_conv_gemm` take the pre-split LDS-DMA path for this GEMM if its A operand had a
        shadow?  (Shape question only: asked with dummy aligned pointers.)"""
        if not self.split_mode:
            return False
        d = L.ConvDesc()
        P_ = 1 << 20
        d.a, d.lda, d.a_img_stride = P_, cin, h * w
        d.a_split, d.ldas = P_, 2 * cin
        d.w, d.out, d.ldc, d.o_img_stride = P_, P_, (cout if ldc is None else ldc), h * w
        if has_res:
            d.res, d.ldr, d.r_img_stride = P_, (cout if ldr is None else ldr), h * w
        d.n_img, d.H, d.W, d.Cin, d.Cout, d.taps = n_img, h, w, cin, cout, taps
        d.prec, d.act, d.epi, d.tile = self.prec, act, L.EPI_AFFINE, 0
        return int(self.lib.uavsal_conv_uses_split(C.byref(d))) == 1

    def _scr(self, kind, n, h, w, c) -> V:
        """Scratch for the expanded tensors of an inverted-residual block and the Winograd planes.  With the arena: an
        anonymous buffer of the pool, live from its producer to its last reader (blocks that run back to back end up on the
        same addresses, concurrent lanes never do).  Without: one pool per (kind, lane)."""
        numel = n * h * w * c
        if self.use_arena:
            self._scr_serial += 1
            return V(self._ref((kind, self._scr_serial), numel), n, h, w, c)
        key = (kind, self._lane)
        if self._dry:
            self._scratch_need[key] = max(self._scratch_need.get(key, 0), numel)
            return V(_Fake(), n, h, w, c)
        return V(self._scratch[key], n, h, w, c)

    # ---- parallel branches (uavsal_plan lanes) -----------------------------------------
    def fork(self, lane):
        """Following ops (until `main()`) go to `lane`, which starts after everything recorded on
        lane 0 so far."""
        self._meta(kind="sync", name="fork%d" % lane, flops=0.0, bytes=0.0)
        self._lane_open.setdefault(lane, self._lop)
        if not self._dry:
            r = self.lib.uavsal_plan_add_fork(self.plan, lane)
            if r < 0:
                L.check(r, "plan_add_fork")
            L.check(self.lib.uavsal_plan_set_lane(self.plan, lane), "plan_set_lane")
        self._lane = lane

    def main(self):
        if not self._dry:
            L.check(self.lib.uavsal_plan_set_lane(self.plan, 0), "plan_set_lane")
        self._lane = 0

    def join(self, lane):
        self._meta(kind="sync", name="join%d" % lane, flops=0.0, bytes=0.0)
        for r_ in self._lane_refs.pop(lane, ()):          # what ran on the lane may have been running until here
            r_.last = max(r_.last, self._lop)
        self._lane_open.pop(lane, None)
        if not self._dry:
            r = self.lib.uavsal_plan_add_join(self.plan, lane)
            if r < 0:
                L.check(r, "plan_add_join")

    def _dev(self, t: torch.Tensor) -> torch.Tensor:
        return t.contiguous().to(self.device)      # kept alive by the weight cache

    def _affine(self, bn, cout):
        if isinstance(bn, (list, tuple)):        # several convs of the same input as ONE GEMM: outputs side by side
            key = ("bn",) + tuple(id(b_) for b_ in bn) + (cout,)
            if key not in self._wcache:
                parts = [P.fold_bn(b_) for b_ in bn]
                s, b = torch.cat([p_[0] for p_ in parts]), torch.cat([p_[1] for p_ in parts])
                n = P.roundup(cout, 32)
                self._wcache[key] = (self._dev(P.pad_vec(s, n, 1.0)), self._dev(P.pad_vec(b, n, 0.0)))
            return self._wcache[key]
        key = ("bn", id(bn), cout)
        if key not in self._wcache:
            s, b = P.fold_bn(bn)
            n = P.roundup(cout, 32)
            self._wcache[key] = (self._dev(P.pad_vec(s, n, 1.0)), self._dev(P.pad_vec(b, n, 0.0)))
        return self._wcache[key]

    def _convw(self, conv, sl=None, gate_interleave=0, natural=False, dwproj=False, k32=False, prec_name=None):
        """`natural`: the pre-split LDS-DMA path takes the weights as [K step][Cout][hi 32 | lo 32] ('f16x3i');
        `dwproj`: the split-fp16 depthwise -> projection kernel takes [K step of 16][Cout][hi 16 | lo 16] ('f16x3j');
        `k32`: fp32 kernels with 32-float K stages (tiles 8 / 9; differs from 'f32' for 3x3 weights only)."""
        prec_name = prec_name or self.prec_name
        layout = "f16x3i" if natural else ("f16x3j" if dwproj and prec_name == "f16x3" else prec_name)
        multi = isinstance(conv, (list, tuple))
        if k32 and layout == "f32" and (conv[0] if multi else conv).weight.shape[-1] == 3:
            layout = "f32k32"
        key = ("w",) + (tuple(id(c_) for c_ in conv) if multi else (id(conv),)) + (sl, layout, gate_interleave)
        if key not in self._wcache:
            w = torch.cat([c_.weight.detach() for c_ in conv], 0) if multi else conv.weight.detach()
            if sl is not None:
                w = w[:, sl[0]:sl[1]]
            if gate_interleave:      # ConvLSTM: row g*hid + c  ->  4*c + g  (gates i,f,o,g adjacent)
                hid = gate_interleave
                w = w.reshape(4, hid, *w.shape[1:]).permute(1, 0, 2, 3, 4).reshape(4 * hid, *w.shape[1:])
            self._wcache[key] = self._dev(P.pack_conv_weight(w, layout))
        return self._wcache[key]

    def _prec_for(self, name) -> str:
        best, val = -1, self.prec_name
        for k, v in self.prec_overrides.items():
            if name.startswith(k) and len(k) > best:
                best, val = len(k), v
        return val

    def _tile_of(self, n_img, h, w, cout, epi) -> int:
        d = L.ConvDesc()
        d.n_img, d.H, d.W, d.Cout, d.prec, d.epi, d.tile = n_img, h, w, cout, self.prec, epi, 0
        d.out = 1 << 20
        return int(self.lib.uavsal_conv_tile(C.byref(d)))

    # ------------------------------------------------------------------ op recorders
    def _meta(self, **kw):
        self._flush_poison()
        self._lop += 1
        self._op_idx[kw.get("name")] = len(self.ops_meta)       # == index of the op in the native plan
        self.ops_meta.append(kw)

    def _patch(self, name, slot, ptr):
        L.check(self.lib.uavsal_plan_patch_ptr(self.plan, self._op_idx[name], slot, ptr), "plan_patch_ptr(%s)" % name)

    def _add(self, fn, desc, what):
        r = fn(self.plan, C.byref(desc))
        if r < 0:
            L.check(r, what)

    def conv(self, name, a: V, conv, bn, out: V, act, taps=1, res: Optional[V] = None, wslice=None,
             epi=L.EPI_AFFINE, aux: Optional[V] = None, n_img=None, strides=None, cout=None,
             out2: Optional[V] = None, gate_interleave=0, dw=None, n_group=0):
        """`dw=(dw_conv, dw_bn, stride)`: `a` is the expanded tensor and the depthwise 3x3 + BN + ReLU6
        is produced inside this GEMM's loader (uavsal_conv_desc.dw_*).
        `n_group`: `conv` / `bn` are lists of 1x1 convs with `n_group` outputs each whose inputs lie side by side in `a`'s rows
        (a = the first one's view): one launch (uavsal_conv_desc.n_group / a_group_off)."""
        cin = a.c
        cout = out.c if cout is None else cout
        n_img = a.n if n_img is None else n_img
        hin, win = a.h, a.w
        if dw is not None:
            a = V(a.t, a.n, (hin - 1) // dw[2] + 1, (win - 1) // dw[2] + 1, a.c, a.ld, a.coff)
        hw = a.h * a.w
        flops = 2.0 * n_img * hw * cin * cout * taps
        byts = 4.0 * n_img * hw * (cin + cout) + 4.0 * cin * cout * taps
        if dw is not None:      # the launch also does the depthwise: reads E (hin x win), D never exists
            flops += 18.0 * n_img * hw * cin
            byts = 4.0 * n_img * (hin * win * cin + hw * cout) + 4.0 * cin * (cout + 11)
        self._meta(kind="conv%d" % (3 if taps == 9 else 1), name=name, flops=flops, bytes=byts,
                   M=n_img * hw, K=cin * taps, Nc=cout)
        self._touch(a, out, res, aux, out2)
        # split shadows (f16x3): can this launch write one for its output / read its input pre-split?
        shadow_out = False
        if self.split_mode:
            aligned = cout % 4 == 0 and out.ld % 4 == 0 and (res is None or res.ld % 4 == 0)
            if epi == L.EPI_AFFINE:
                shadow_out = aligned and act != L.ACT_SIGMOID
            elif epi == L.EPI_TWA:       # the vector ConvTWA update only exists in the 1x1-fragment tiles
                shadow_out = aligned and self._tile_of(n_img, a.h, a.w, cout, epi) in (3, 4)
            if self._dry:
                if not shadow_out and out.key is not None:
                    self._no_shadow.add(out.key)
                if (a.key is not None and a.key not in self._no_shadow and dw is None and strides is None
                        and epi == L.EPI_AFFINE and a.ld % 32 == 0 and (a.coff % a.ld) % 32 == 0 and self._would_split(
                            n_img, a.h, a.w, cin, cout, taps, act, res is not None, out.ld, res.ld if res is not None else None)):
                    self._split_want.add(a.key)
        if self._dry:
            return
        d = L.ConvDesc()
        st = strides or {}
        d.a, d.lda, d.a_img_stride = a.ptr, a.ld, st.get("a", hin * win if dw is not None else hw)
        if a.sp is not None and dw is None and strides is None and epi == L.EPI_AFFINE:
            d.a_split, d.ldas = a.sp_ptr, 2 * a.ld
        if out.sp is not None and shadow_out:
            d.out_split, d.ldos = out.sp_ptr, 2 * out.ld
        if dw is not None:
            key = ("dw", id(dw[0]))
            if key not in self._wcache:
                s_, b_ = P.fold_bn(dw[1])
                self._wcache[key] = (self._dev(P.pack_dw_weight(dw[0].weight)), self._dev(s_), self._dev(b_))
            w9, s_, b_ = self._wcache[key]
            d.dw_w9c, d.dw_scale, d.dw_bias = w9.data_ptr(), s_.data_ptr(), b_.data_ptr()
            d.dw_stride, d.dw_Hin, d.dw_Win = dw[2], hin, win
            self.ops_meta[-1]["fused_dw"] = True
        if bn is not None:
            s, b = self._affine(bn, cout)
            d.scale, d.bias = s.data_ptr(), b.data_ptr()
        else:
            d.scale, d.bias = None, None
        d.out, d.ldc, d.o_img_stride = out.ptr, out.ld, st.get("o", hw)
        if res is not None:
            d.res, d.ldr, d.r_img_stride = res.ptr, res.ld, st.get("r", hw)
        else:
            d.res, d.ldr, d.r_img_stride = None, 0, hw
        if aux is not None:
            d.aux, d.ldx, d.x_img_stride = aux.ptr, aux.ld, st.get("x", hw)
        else:
            d.aux, d.ldx, d.x_img_stride = None, 0, hw
        d.n_img, d.H, d.W = n_img, a.h, a.w
        d.Cin, d.Cout, d.taps = cin, cout, taps
        pn = self._prec_for(name)
        d.prec, d.act, d.epi, d.tile = L.PREC[pn], act, epi, 0
        d.n_group, d.a_group_off = n_group, (cin if n_group else 0)
        # GEMMs on a side lane run beside grid-filling GEMMs of the main lane: the 64 x 64 instance with 32-float K stages
        # needs 32 KB of LDS and 122 VGPRs, so one of its workgroups fits on a CU next to two of the main lane's
        # (64 KB, 155 VGPRs each) instead of waiting for them to retire
        side_tile = int(os.environ.get("UAVSAL_SIDE_TILE", "11"))      # (5.155 vs 5.17 ms per step, same box, two runs each)
        if side_tile and self._lane != 0 and pn == "f32" and epi == L.EPI_AFFINE and dw is None and cin % 32 == 0:
            d.tile = side_tile
        if name in _TILE_OVERRIDE:           # experiments: UAVSAL_TILE_OVERRIDE="ctx.0.pw=11,ctx.1.pl=11"
            d.tile = _TILE_OVERRIDE[name]
        if out2 is not None:
            d.out2, d.ld2 = out2.ptr, out2.ld
        if self.stream_k:
            # one workspace per lane: launches on a lane are ordered on one stream (uavsal_conv_desc.sk_ws)
            ws = self._sk_ws.get(self._lane)
            if ws is None:
                ws = self._sk_ws[self._lane] = torch.zeros(int(self.lib.uavsal_streamk_workspace_bytes()),
                                                           dtype=torch.uint8, device=self.device)
            d.sk_ws, d.sk_ws_bytes = ws.data_ptr(), ws.numel()
        d.err = self._err
        d.sk_spin_limit, d.sk_debug_drop = self._sk_debug      # test hooks (model._sk_debug), normally (0, 0)
        # weights last: their 16-bit packing depends on which kernel the descriptor selects
        d.w = 1 << 20
        split = int(self.lib.uavsal_conv_uses_split(C.byref(d))) == 1
        if a.t is None and not split:
            raise RuntimeError("%s: its input only exists as a split shadow but the GEMM is not eligible" % name)
        dwproj = int(self.lib.uavsal_conv_dwproj(C.byref(d)))
        tile = int(self.lib.uavsal_conv_tile(C.byref(d)))
        d.w = self._convw(conv, wslice, gate_interleave, natural=split, dwproj=dwproj != 0, k32=tile in (8, 9, 10, 11),
                          prec_name=pn).data_ptr()
        self.ops_meta[-1]["prec"] = pn
        self.ops_meta[-1]["split"] = split
        self.ops_meta[-1]["tile"] = tile
        self.ops_meta[-1]["streamk"] = int(self.lib.uavsal_conv_streamk_grid(C.byref(d)))
        self.ops_meta[-1]["dwproj"] = dwproj
        self._add(self.lib.uavsal_plan_add_conv, d, "plan_add_conv(%s)" % name)

    def conv3_wino(self, name, a: V, conv, bn, out: V, act, wslice=None, n_img=None, strides=None, twa=None, gemm_tile=0, r=2,
                   segs=None):
        """Dense 3x3 conv (stride 1, padding 1) as Winograd F(r x r, 3x3), exact-fp32 mode only: input transform, ONE GEMM
        launch over the (r + 2)^2 transform planes (per-plane weights), output transform with the epilogue -- 2.25x (r = 2)
        or 4x (r = 4) fewer MFMA FLOPs than the implicit GEMM (csrc/winograd.hip).  `twa=(x_t, pre_t)`: ConvTWA update in the output transform.
        `segs`: the input is the channel concatenation of these views, each resized to `a`'s map size inside the input transform
        when it lives on a smaller map (uavsal_wino_desc.n_seg); `a` then only carries the shape (its `t` is None)."""
        cin, cout = a.c, out.c
        n = a.n if n_img is None else n_img
        hw = a.h * a.w
        tiles = n * ((a.h + r - 1) // r) * ((a.w + r - 1) // r)
        pp = (r + 2) * (r + 2)
        mp = P.roundup(tiles, 128)
        st = strides or {}
        v = self._scr("WV", pp, mp, 1, cin)
        mm = self._scr("WM", pp, mp, 1, cout)
        in_bytes = 4.0 * (sum(sg.n * sg.h * sg.w * sg.c for sg in segs) if segs else n * hw * cin)
        self._meta(kind="wino_in", name=name + ".xin", flops=0.0, bytes=in_bytes + 4.0 * float(pp) * tiles * cin)
        self._touch(a, v, *(segs or ()))
        if not self._dry:
            wi = L.WinoDesc()
            if segs:
                assert sum(sg.c for sg in segs) == cin and len(segs) <= 3 and not strides
                wi.n_seg = len(segs)
                for i, sg in enumerate(segs):
                    wi.seg_in[i], wi.seg_ld[i], wi.seg_c[i], wi.seg_H[i], wi.seg_W[i] = sg.ptr, sg.ld, sg.c, sg.h, sg.w
            else:
                wi.inp, wi.ldi, wi.in_img_stride = a.ptr, a.ld, st.get("a", hw)
            wi.out, wi.ldo = v.ptr, cin
            wi.n_img, wi.H, wi.W, wi.C, wi.Mp, wi.R = n, a.h, a.w, cin, mp, r
            self._add(self.lib.uavsal_plan_add_wino_input, wi, "plan_add_wino_input(%s)" % name)
        # WINO_TAIL_PLANES (experiment): the planes that only fill the GEMM's last, partial round of 128 x 128 tiles (36 planes x
        # 15 x 2 tiles on 512 resident workgroups = 2.1 rounds) as a second launch on 64 x 64 tiles
        tail = WINO_TAIL_PLANES if (gemm_tile in (0, 8) and twa is None and pp > WINO_TAIL_PLANES > 0) else 0
        plane_parts = [(0, pp - tail, gemm_tile, name)] + ([(pp - tail, tail, 11, name + ".tail")] if tail else [])
        if not self._dry:
            key = ("wino", id(conv), wslice, r)
            if key not in self._wcache:
                w = conv.weight.detach()
                if wslice is not None:
                    w = w[:, wslice[0]:wslice[1]]
                self._wcache[key] = self._dev(P.pack_wino_weight(w, r))
        wgs = P.roundup(cout, 32) * P.roundup(cin, 32)
        for (p0, pn, tile_, nm_) in plane_parts:
            self._meta(kind="conv1", name=nm_, flops=2.0 * pn * tiles * cin * cout,
                       bytes=4.0 * pn * (tiles * (cin + cout) + cin * cout), M=pn * mp, K=cin, Nc=cout,
                       direct_flops=2.0 * n * hw * cin * cout * 9 * pn / pp)
            self._touch(v, mm)
            if self._dry:
                continue
            d = L.ConvDesc()
            d.a, d.lda, d.a_img_stride = v.ptr + 4 * p0 * mp * cin, cin, mp
            d.w, d.w_group_stride = self._wcache[key].data_ptr() + 4 * p0 * wgs, wgs
            d.out, d.ldc, d.o_img_stride = mm.ptr + 4 * p0 * mp * cout, cout, mp
            d.n_img, d.H, d.W, d.Cin, d.Cout, d.taps = pn, mp, 1, cin, cout, 1
            d.prec, d.act, d.epi, d.tile = L.PREC["f32"], L.ACT_NONE, L.EPI_AFFINE, tile_      # Winograd plans are exact fp32
            d.err = self._err
            m_ = self.ops_meta[-1]
            m_["split"], m_["tile"], m_["streamk"], m_["dwproj"] = False, int(self.lib.uavsal_conv_tile(C.byref(d))), 0, 0
            m_["prec"] = "f32"
            self._add(self.lib.uavsal_plan_add_conv, d, "plan_add_conv(%s)" % nm_)
        self._meta(kind="wino_out", name=name + ".xout", flops=0.0, bytes=4.0 * (float(pp) * tiles * cout + n * hw * cout))
        self._touch(mm, out, *(twa or ()), *((a,) if twa is not None else ()))
        if out.key is not None and self._dry:
            self._no_shadow.add(out.key)            # the output transform does not write split shadows
        if not self._dry:
            wo = L.WinoDesc()
            wo.inp, wo.ldi = mm.ptr, cout
            wo.out, wo.ldo, wo.out_img_stride = out.ptr, out.ld, st.get("o", hw)
            wo.n_img, wo.H, wo.W, wo.C, wo.Mp, wo.R = n, a.h, a.w, cout, mp, r
            if bn is not None:
                s_, b_ = self._affine(bn, cout)
                wo.scale, wo.bias = s_.data_ptr(), b_.data_ptr()
            wo.act, wo.epi = act, L.EPI_AFFINE
            if twa is not None:
                xt, pre = twa
                wo.epi = L.EPI_TWA
                wo.res, wo.ldr, wo.res_img_stride = xt.ptr, xt.ld, st.get("r", hw)
                wo.aux, wo.ldx, wo.aux_img_stride = pre.ptr, pre.ld, st.get("x", hw)
                wo.hprev, wo.ldh, wo.h_img_stride = a.ptr, a.ld, st.get("a", hw)
            self._add(self.lib.uavsal_plan_add_wino_output, wo, "plan_add_wino_output(%s)" % name)

    def dw(self, name, a: V, conv, bn, out: V, stride, dilation):
        c = a.c
        ho, wo = (a.h - 1) // stride + 1, (a.w - 1) // stride + 1
        byts = 4.0 * a.n * c * (a.h * a.w + ho * wo) + 4.0 * 9 * c + 4.0 * 2 * c
        self._meta(kind="dw", name=name, flops=2.0 * 9 * a.n * ho * wo * c, bytes=byts, stride=stride,
                   dil=dilation if not isinstance(dilation, (list, tuple)) else tuple(dilation), patches44=a.n * ((ho + 3) // 4) * ((wo + 3) // 4) * (c // 4))
        self._touch(a, out)
        if self._dry:
            return
        grouped = isinstance(conv, (list, tuple))      # several dilated branches of one map: channel groups with their own dilation
        key = ("dw",) + tuple(id(c_) for c_ in conv) if grouped else ("dw", id(conv))
        if key not in self._wcache:
            if grouped:
                parts = [P.fold_bn(b_) for b_ in bn]
                self._wcache[key] = (self._dev(torch.cat([P.pack_dw_weight(c_.weight) for c_ in conv], 1)),
                                     self._dev(torch.cat([p_[0] for p_ in parts])), self._dev(torch.cat([p_[1] for p_ in parts])))
            else:
                s, b = P.fold_bn(bn)
                self._wcache[key] = (self._dev(P.pack_dw_weight(conv.weight)), self._dev(s), self._dev(b))
        w9, s, b = self._wcache[key]
        d = L.DwDesc()
        if grouped:
            d.dil_group_c = c // len(conv)
            for gi, dl in enumerate(dilation):
                d.dil_groups[gi] = dl
            dilation = dilation[0]
        d.inp, d.ldi = a.ptr, a.ld
        d.w9c, d.scale, d.bias = w9.data_ptr(), s.data_ptr(), b.data_ptr()
        if out.t is None:            # the projection GEMM stages this tensor pre-split: no fp32 copy
            d.out, d.ldo = None, out.ld
            d.out_split, d.ldos = out.sp_ptr, 2 * out.ld
            self.ops_meta[-1]["split_out"] = True
        else:
            d.out, d.ldo = out.ptr, out.ld
        d.n_img, d.H, d.W, d.C = a.n, a.h, a.w, c
        d.stride, d.dilation, d.act = stride, dilation, L.ACT_RELU6
        self.ops_meta[-1]["kernel"] = L.DW_KERNEL.get(int(self.lib.uavsal_dw_variant(C.byref(d))), "dw3x3")
        self._add(self.lib.uavsal_plan_add_dw, d, "plan_add_dw(%s)" % name)

    def dw_dot(self, name, a: V, dwc, dwbn, pl, plbn, out: V, act):
        """Depthwise 3x3 + BN + ReLU6 -> projection to ONE channel + BN + act as one bandwidth-bound launch (uavsal_dw3x3_dot):
        the tail of conv_out_st (model.py:333-334, 372-373).  Exact fp32 in every precision mode of the plan."""
        c = a.c
        self._meta(kind="dw_dot", name=name, flops=2.0 * 10 * a.n * a.h * a.w * c, bytes=4.0 * a.n * a.h * a.w * (c + 1) + 4.0 * 12 * c,
                   stride=1, dil=1, kernel="dw3x3_dot_kernel<4, 4>")
        self._touch(a, out)
        if self._dry:
            return
        key = ("dwdot", id(dwc), id(pl))
        if key not in self._wcache:
            s, b = P.fold_bn(dwbn)
            s2, b2 = P.fold_bn(plbn)
            self._wcache[key] = (self._dev(P.pack_dw_weight(dwc.weight)), self._dev(s), self._dev(b),
                                 self._dev(pl.weight.detach().float().reshape(-1)), self._dev(s2.reshape(1)), self._dev(b2.reshape(1)))
        w9, s, b, w2, s2, b2 = self._wcache[key]
        d = L.DwDotDesc()
        d.inp, d.ldi = a.ptr, a.ld
        d.w9c, d.scale, d.bias, d.w2, d.scale2, d.bias2 = (t.data_ptr() for t in (w9, s, b, w2, s2, b2))
        d.out, d.ldo = out.ptr, out.ld
        d.n_img, d.H, d.W, d.C, d.act = a.n, a.h, a.w, c, act
        self._add(self.lib.uavsal_plan_add_dw_dot, d, "plan_add_dw_dot(%s)" % name)

    def bilinear(self, name, a: V, out: V, src_mod=None, src_div=1):
        self._meta(kind="bilinear", name=name, flops=0.0, bytes=4.0 * out.n * out.h * out.w * out.c * 2)
        self._touch(a, out)
        if self._dry:
            return
        d = L.BilinearDesc()
        d.inp, d.ldi, d.Hi, d.Wi = a.ptr, a.ld, a.h, a.w
        d.out, d.ldo, d.Ho, d.Wo = out.ptr, out.ld, out.h, out.w
        d.n_out, d.C = out.n, a.c
        if out.sp is not None:
            d.out_split, d.ldos = out.sp_ptr, 2 * out.ld
        d.src_mod, d.src_div = (out.n if src_mod is None else src_mod), src_div
        self._add(self.lib.uavsal_plan_add_bilinear, d, "plan_add_bilinear(%s)" % name)

    def layout(self, name, src, dst, n, c, hw, ld, to_nhwc, cpad=0):
        """`src` / `dst`: a view (its address is taken once the op is open, so that the arena sees the use at this op) or a raw
        device address (the caller's boundary tensors)."""
        self._meta(kind="layout", name=name, flops=0.0, bytes=8.0 * n * c * hw)
        self._touch(*(t for t in (src, dst) if isinstance(t, V)))
        if self._dry:
            return
        d = L.LayoutDesc()
        src_ptr, dst_ptr = (t.ptr if isinstance(t, V) else t for t in (src, dst))
        d.inp, d.out, d.n_img, d.C, d.HW, d.ld, d.to_nhwc, d.Cpad = src_ptr, dst_ptr, n, c, hw, ld, to_nhwc, cpad
        self._add(self.lib.uavsal_plan_add_layout, d, "plan_add_layout(%s)" % name)

    def fused_block(self, name, x: V, blk, out: V) -> bool:
        """The whole inverted-residual block as ONE launch (uavsal_fused_ir: the expanded tensors stay in LDS),
        where an instance exists -- the bandwidth-bound small-channel blocks features[1..7].  False = not taken."""
        seq = blk.conv
        if not self.fuse_blocks or getattr(blk, "dilation", 1) != 1:
            return False
        d = L.FusedIrDesc()
        d.Cin, d.hidden, d.Cout, d.stride = x.c, blk.hidden, out.c, blk.stride
        d.w1 = (1 << 20) if blk.expand_ratio != 1 else None
        kind = int(self.lib.uavsal_fused_ir_supported(C.byref(d)))
        if not kind:
            return False
        natural = kind == 2          # csrc/fused_mid.hip: 1x1 weights in their own layout
        if natural:
            # one workgroup per 4 x 8 output patch and CU-wide LDS: taken where the launch is about one round of the chip
            # (the 23x40 backbone maps at one clip); bigger launches keep expand GEMM + depthwise / projection launches
            wgs = x.n * ((x.h + 3) // 4) * ((x.w + 7) // 8)
            if not (MID_MIN_WGS <= wgs <= MID_MAX_WGS):
                return False
        ho, wo = (x.h - 1) // blk.stride + 1, (x.w - 1) // blk.stride + 1
        self._meta(kind="fused_ir", name=name, kernel="%s<%d, %d, %d%s>" % ("fused_mid_kernel" if natural else "fused_ir_kernel", x.c, blk.hidden, out.c,
                                                                    "" if natural else ", %d" % blk.stride),
                   flops=2.0 * x.n * ((x.h * x.w * x.c * blk.hidden if blk.expand_ratio != 1 else 0)
                                      + ho * wo * blk.hidden * (9 + out.c)),
                   bytes=4.0 * x.n * (x.h * x.w * x.c + ho * wo * out.c * (2 if blk.use_res_connect else 1)),
                   unfused_bytes=4.0 * x.n * (x.h * x.w * (x.c + (2 * blk.hidden if blk.expand_ratio != 1 else 0))
                                              + ho * wo * (2 * blk.hidden + out.c)),
                   # what the matrix pipe executes in the mid kernel: every 4 x 8 patch expands its whole 6 x 10 halo (64 MFMA
                   # rows) and projects 32 rows, edge patches included
                   **({"flops_executed": 2.0 * wgs * blk.hidden * (64 * x.c + 32 * out.c)} if natural else {}))
        self._touch(x, out)
        if out.key is not None:
            self._no_shadow.add(out.key)            # this kernel does not write split shadows
        if self._dry:
            return True
        if blk.expand_ratio != 1:
            pw, pwbn, dwc, dwbn, pl, plbn = seq[0][0], seq[0][1], seq[1][0], seq[1][1], seq[2], seq[3]
        else:
            pw, pwbn, dwc, dwbn, pl, plbn = None, None, seq[0][0], seq[0][1], seq[1], seq[2]
        key = ("fused", id(dwc), natural)
        if key not in self._wcache:
            ws = {}
            if pw is not None:
                s_, b_ = P.fold_bn(pwbn)
                w1 = pw.weight.detach().float().cpu().reshape(blk.hidden, x.c)
                ws["w1"] = self._dev((w1 if natural else w1.t()).contiguous())
                ws["s1"], ws["b1"] = self._dev(s_), self._dev(b_)
            s_, b_ = P.fold_bn(dwbn)
            ws["wd"], ws["sd"], ws["bd"] = self._dev(P.pack_dw_weight(dwc.weight)), self._dev(s_), self._dev(b_)
            s_, b_ = P.fold_bn(plbn)
            w2 = pl.weight.detach().float().cpu().reshape(out.c, blk.hidden)
            ws["w2"] = self._dev((w2 if natural else w2.t()).contiguous())
            ws["s2"], ws["b2"] = self._dev(s_), self._dev(b_)
            self._wcache[key] = ws
        ws = self._wcache[key]
        d.inp, d.ldi = x.ptr, x.ld
        if pw is not None:
            d.w1, d.scale1, d.bias1 = ws["w1"].data_ptr(), ws["s1"].data_ptr(), ws["b1"].data_ptr()
        d.wd, d.scale_d, d.bias_d = ws["wd"].data_ptr(), ws["sd"].data_ptr(), ws["bd"].data_ptr()
        d.w2, d.scale2, d.bias2 = ws["w2"].data_ptr(), ws["s2"].data_ptr(), ws["b2"].data_ptr()
        if blk.use_res_connect:
            d.res, d.ldr = x.ptr, x.ld
        d.out, d.ldo = out.ptr, out.ld
        d.n_img, d.H, d.W = x.n, x.h, x.w
        self._add(self.lib.uavsal_plan_add_fused_ir, d, "plan_add_fused_ir(%s)" % name)
        return True

    def ir_block(self, name, x: V, blk, out: V, final_act=L.ACT_NONE, expanded: Optional[V] = None):
        """pw-expand + BN + ReLU6 -> dw3x3 + BN + ReLU6 -> pw-linear + BN [+ x]
        (dwBlock, reference model.py:74-103; torchvision InvertedResidual).
        `expanded`: the block's expanded tensor already exists (several blocks' expands run as one GEMM)."""
        if expanded is None and final_act == L.ACT_NONE and self.fused_block(name, x, blk, out):
            return
        seq = blk.conv
        stride, dil = blk.stride, getattr(blk, "dilation", 1)
        # a block whose expanded tensor would be bigger than BLOCK_CHUNK_BYTES runs in chunks of whole frames (the block is
        # per-frame arithmetic; the launches stay many rounds of the chip): the arena's peak is set by the biggest E, not by
        # the layer count -- 720x1280 x 64 frames: fucbst's 7.1 GB E in two halves, peak 16.97 -> ~13 GB
        if (expanded is None and blk.expand_ratio != 1 and x.n > 1 and self.use_arena and out.c > 1      # (the decoder's map is bound per call by op name)
                and 4 * x.n * x.h * x.w * blk.hidden > BLOCK_CHUNK_BYTES):
            per = 4 * x.h * x.w * blk.hidden
            step = max(1, BLOCK_CHUNK_BYTES // per)
            nchunk = (x.n + step - 1) // step
            step = (x.n + nchunk - 1) // nchunk                     # equal chunks
            for ci, f0 in enumerate(range(0, x.n, step)):
                cnt = min(step, x.n - f0)
                self._ir_block_one("%s#%d" % (name, ci) if nchunk > 1 else name, x.frames(f0, cnt), blk, out.frames(f0, cnt), final_act)
            return
        self._ir_block_one(name, x, blk, out, final_act, expanded)

    def _ir_block_one(self, name, x: V, blk, out: V, final_act=L.ACT_NONE, expanded: Optional[V] = None):
        seq = blk.conv
        stride, dil = blk.stride, getattr(blk, "dilation", 1)
        if blk.expand_ratio != 1:
            if expanded is not None:
                e = expanded
            else:
                e = self._scr("E", x.n, x.h, x.w, blk.hidden)
                self.conv(name + ".pw", x, seq[0][0], seq[0][1], e, L.ACT_RELU6)
            dwc, dwbn, pl, plbn = seq[1][0], seq[1][1], seq[2], seq[3]
        else:
            e = x
            dwc, dwbn, pl, plbn = seq[0][0], seq[0][1], seq[1], seq[2]
        ho, wo = (x.h - 1) // stride + 1, (x.w - 1) // stride + 1
        if (DW_DOT and out.c == 1 and stride == 1 and dil == 1 and blk.expand_ratio != 1 and not blk.use_res_connect
                and blk.hidden % 256 == 0 and blk.hidden <= 2048 and self.fuse_dw is not False):
            self.dw_dot(name + ".dwpl", e, dwc, dwbn, pl, plbn, out, final_act)      # a dot product per pixel: bandwidth-bound
            return
        if dil == 1 and blk.expand_ratio != 1 and (self.fuse_dw or (
                self.fuse_dw is None and self._prec_for(name + ".dwpl") in ("f32", "f16x3") and stride == 1 and blk.hidden % 16 == 0
                and x.n * x.h * x.w * blk.hidden >= FUSE_DW_MIN_WORK and _dwproj_patch_waste(x.h, x.w) <= FUSE_DW_MAX_WASTE)):
            # depthwise computed inside the projection GEMM's loader: D never reaches HBM
            self.conv(name + ".dwpl", e, pl, plbn, out, final_act, res=x if blk.use_res_connect else None,
                      dw=(dwc, dwbn, stride))
            return
        res = x if blk.use_res_connect else None
        if dil == 1 and self._would_split(x.n, ho, wo, blk.hidden, out.c, 1, final_act, res is not None, out.ld,
                                          res.ld if res is not None else None):
            dd = self._scr_split(x.n, ho, wo, blk.hidden)      # D only ever exists as hi/lo fp16 planes
        else:
            dd = self._scr("D", x.n, ho, wo, blk.hidden)
        self.dw(name + ".dw", e, dwc, dwbn, dd, stride, dil)
        self.conv(name + ".pl", dd, pl, plbn, out, final_act, res=res)

    def _mark(self, stage, start):
        self.stage_ranges[stage] = (start, len(self.ops_meta))

    # ------------------------------------------------------------------ the forward, recorded
    def _build(self):
        m, N, h, w = self.model, self.N, self.h, self.w
        hw = h * w
        R6, NONE = L.ACT_RELU6, L.ACT_NONE
        dev = self.device
        if not self._dry:
            # boundary staging (NCHW, as the reference caller hands them over)
            # (launch-loop mode binds the caller's tensors into the plan before every run: the staging tensors of the frames and
            # priors are then shapes only -- zero-stride views of one 4 KB block, not 0.8 GB at 64 frames of 720x1280 -- and
            # `launch` refuses to run a plan that was never bound)
            def _stage(shape, dtype):
                if self.inplace:
                    return torch.zeros(1024, dtype=dtype, device=dev)[:1].expand(shape)
                return torch.empty(shape, dtype=dtype, device=dev)
            self.x_in = _stage((N, 3, self.H, self.W), self.in_dtype)
            self.cb0_in = _stage((1 if self.static_priors else N, 8, h, w), torch.float32)
            self.cb1_in = _stage((1 if self.static_priors else N, 20, h, w), torch.float32)
            self._bound = False
            self.state_in = torch.zeros((self.n_seq, 256, h, w), dtype=torch.float32, device=dev)
            self.zero_state = torch.zeros((self.n_seq, 256, h, w), dtype=torch.float32, device=dev)    # never written
            self.state_out = torch.empty((self.n_seq, 256, h, w), dtype=torch.float32, device=dev)
            self.cstate_in = torch.zeros((self.n_seq, 256, h, w), dtype=torch.float32, device=dev)
            self.cstate_out = torch.empty((self.n_seq, 256, h, w), dtype=torch.float32, device=dev)
            self.out = torch.empty((N, hw), dtype=torch.float32, device=dev)
            self.logits = torch.empty((N, hw), dtype=torch.float32, device=dev) if self.keep_taps else None
        feats = m.sfnet.features.features

        # ---- boundary: state and priors NCHW -> NHWC
        s0 = len(self.ops_meta)
        h0 = self._buf("h0", self.n_seq, h, w, 256, pinned=self.persistent)
        # buffers written by kernels that do not produce split shadows
        self._no_shadow.update(("h0", "c0", "gauss_in", "ob_in", "f0", "ctx_sum", "lstm_pre", "lstm_c", "twa_pre"))
        self._no_shadow.update("st%d_dif" % i for i in range(len(m.st_layer)))
        lstm_model = getattr(m, "rnn_type", "twa") == "lstm"
        c0 = self._buf("c0", self.n_seq, h, w, 256, pinned=self.persistent) if lstm_model else None
        Np = 1 if self.static_priors else N
        # which priors this model has (reference model.py:281-324: a disabled prior has no net, and with none at all the two
        # fusion blocks do not exist either); enabled priors keep the reference's concat order gauss | observed | context
        use_g, use_o, use_c = (bool(getattr(m, a, 1)) for a in ("use_gauss_prior", "use_ob_prior", "use_context_prior"))
        num_cb = int(use_g) + int(use_o) + int(use_c)
        cb_off = {}
        for nm_, on_ in (("gauss", use_g), ("ob", use_o), ("ctx", use_c)):
            if on_:
                cb_off[nm_] = 64 * len(cb_off)
        self.use_priors = (use_g, use_o, use_c)
        g0 = self._buf("gauss_in", Np, h, w, 8) if use_g else None
        o0 = self._buf("ob_in", Np, h, w, 20) if use_o else None
        if not self.persistent:          # persistent mode: h0 / c0 ARE the state, staged only on demand (run())
            names = ["state.in"] + (["cstate.in"] if lstm_model else [])
            for nm, src, dst in zip(names, ("state_in", "cstate_in"), (h0, c0)):
                self.layout(nm, None if self._dry else getattr(self, src).data_ptr(), dst, self.n_seq, 256, hw, 256, 1)
        self._mark("boundary_in", s0)

        # ---- backbone: MobileNetV2 features[0:18] (model_feature.py:62-69)
        s0 = len(self.ops_meta)
        H1, W1 = _down(self.H), _down(self.W)
        x = self._buf("f0", N, H1, W1, 32)
        self._meta(kind="stem", name="features.0", flops=2.0 * 27 * 32 * N * H1 * W1,
                   bytes=(4.0 if self.in_dtype == torch.float32 else 1.0) * N * 3 * self.H * self.W + 4.0 * N * H1 * W1 * 32)
        self._touch(x)
        if not self._dry:
            conv0, bn0 = feats[0][0], feats[0][1]
            key = ("stem", id(conv0))
            if key not in self._wcache:
                s, b = P.fold_bn(bn0)
                self._wcache[key] = (self._dev(P.pack_stem_weight(conv0.weight)), self._dev(s), self._dev(b))
            ws, ss, bs = self._wcache[key]
            d = L.StemDesc()
            if self.in_dtype == torch.uint8:
                d.inp, d.in_u8 = None, self.x_in.data_ptr()
            else:
                d.inp, d.in_u8 = self.x_in.data_ptr(), None
            d.w, d.scale, d.bias = ws.data_ptr(), ss.data_ptr(), bs.data_ptr()
            d.out, d.ldo = x.ptr, 32
            d.n_img, d.H, d.W = N, self.H, self.W
            for i in range(3):
                d.mean[i], d.stdv[i] = synth.IMAGENET_MEAN[i], synth.IMAGENET_STD[i]
            self._add(self.lib.uavsal_plan_add_stem, d, "plan_add_stem")
        tapsrc = {}
        cb = None
        # where the prior nets' side lane forks off: beside features.11-17 while those launches are latency-bound (one round of the
        # chip each: up to two clips of 8 frames; 4.26 -> 4.25 ms at one clip), beside features.5-10 from there on (8 clips: 27.92
        # vs 27.97 ms).  A function of the frame count only
        priors_at = int(os.environ.get("UAVSAL_PRIORS_AT", "11" if N <= 16 else "5"))
        tail_halves = TAIL_SPLIT and N >= 2 and N <= TAIL_SPLIT_MAX_FRAMES
        tail_chain = []
        for i in range(1, 18):
            if i == priors_at:
                self._mark("backbone.0-%d" % (priors_at - 1), s0)
                # ---- gaussian / observed prior nets (model.py:349,352): they depend only on the caller's
                #      priors and are needed at fucb_layer, so they run on lanes 1 and 2 beside the backbone.
                #      Recorded HERE, not at the top of the plan: the host launches in recording order, and
                #      with these 14 small launches (+ 4 event operations) in front of it the stem reached
                #      the GPU ~100 us late on every call (rocprofv3 kernel trace, profiles/r2_step_timeline.md).
                s0 = len(self.ops_meta)
                cb = self._buf("cb192", N, h, w, 64 * num_cb) if num_cb else None
                cbs = self._buf("cb_static", 1, h, w, 128) if self.static_priors else cb
                self._no_shadow.add("cb_static")
                forked = set()
                for lane, nm, src, dst, c, on in (
                        (1, "gauss", "cb0_in", g0, 8, use_g),
                        (PRIORS_OB_LANE, "ob", "cb1_in", o0, 20, use_o)):
                    if not on:
                        continue
                    blocks = m.gauss_cb_layer if nm == "gauss" else m.ob_cb_layer
                    mid = self._buf(nm + "1", Np, h, w, 64)
                    sl = cb_off[nm]
                    if lane not in forked:               # (both nets on lane 1: one fork / join pair)
                        self.fork(lane)
                        forked.add(lane)
                    else:
                        L.check(self.lib.uavsal_plan_set_lane(self.plan, lane), "plan_set_lane") if not self._dry else None
                        self._lane = lane
                    self.layout(nm + ".in", None if self._dry else getattr(self, src).data_ptr(), dst, Np, c, hw, c, 1)
                    self.ir_block(nm + ".0", dst, blocks[0], mid)
                    self.ir_block(nm + ".1", mid, blocks[1], cbs.slice(sl, 64))
                    if self.static_priors:      # frame 0 of the net's output -> every frame (same-size resize: an exact copy)
                        self.bilinear(nm + ".bcast", cbs.slice(sl, 64), cb.slice(sl, 64), src_mod=1)
                    self.main()
                self._prior_lanes = sorted(forked)
                self._mark("priors_side", s0)
                s0 = len(self.ops_meta)
            blk = feats[i]
            ho, wo = (x.h - 1) // blk.stride + 1, (x.w - 1) // blk.stride + 1
            y = self._buf("f%d" % i, N, ho, wo, blk.cout)
            if tail_halves and i >= TAIL_SPLIT_FROM:
                # latency-bound launches on the 1/32-scale map (each far below one round of the chip): the two halves of the
                # frames run as two independent chains, the first on lane 2, the second here (blocks are per-frame arithmetic)
                tail_chain.append((i, x, blk, y))
            else:
                self.ir_block("features.%d" % i, x, blk, y)
            x = y
            tapsrc[i] = y
        if tail_chain:
            n0 = N // 2
            self.fork(2)
            for (i, xi, blk, yi) in tail_chain:
                self.ir_block("features.%d/a" % i, xi.frames(0, n0), blk, yi.frames(0, n0))
            self.main()
            for (i, xi, blk, yi) in tail_chain:
                self.ir_block("features.%d/b" % i, xi.frames(n0, N - n0), blk, yi.frames(n0, N - n0))
            self.join(2)
        c3, c4, c5 = tapsrc[6], tapsrc[13], tapsrc[17]
        self.named.update(c3=c3, c4=c4, c5=c5)
        self._mark("backbone.%d-17" % priors_at, s0)

        # ---- SRF-Net head (model.py:139-158)
        s0 = len(self.ops_meta)
        sf = m.sfnet
        aspp = self._buf("aspp", N, c5.h, c5.w, 1024)
        # the four ASPP branches and the two lateral convs are independent small launches on the
        # 1/32 and 1/16 scale maps: spread them over lanes so they fill the chip together
        x5 = self._buf("x5", N, c5.h, c5.w, 256)
        x4 = self._buf("x4", N, c4.h, c4.w, 128)
        # conv_last reads cat[interpolate(x5), interpolate(x4), conv_lv3(c3)] (model.py:151-156).  Winograd plans: the input
        # transform reads the three tensors itself and resizes the two small ones on the fly (uavsal_wino_desc.n_seg) -- no resize
        # launches, no concat buffer (WINO_SEG = 0: the round-4 form; 1: conv_lv4 / conv_lv3 stay on their side lane)
        wino_last = self.winograd and self._prec_for("conv_last") == "f32"
        seg_mode = WINO_SEG if wino_last else 0
        cat = self._buf("srf_cat", N, h, w, 448) if not seg_mode else None
        lv3 = self._buf("lv3", N, h, w, 64) if seg_mode else cat.slice(384, 64)
        branches = (sf.lv5_aspp2, sf.lv5_aspp3, sf.lv5_aspp4)
        aspp_lanes = int(os.environ.get("UAVSAL_ASPP_LANES", "1"))
        fork = self.fork if aspp_lanes else (lambda lane: None)
        join = self.join if aspp_lanes else (lambda lane: None)
        aspp_dw_merged = False
        if int(os.environ.get("UAVSAL_ASPP_MERGE", "1")) and all(b.expand_ratio != 1 for b in branches):
            # the three dilated branches expand the SAME map with the same shape: one GEMM with their output channels
            # side by side (320 -> 3 x 1920: 675 tiles instead of three launches of 225 fighting for the chip on three
            # lanes), then every branch's depthwise + projection on its own lane, reading its slice
            hid = branches[0].hidden
            e3 = self._scr("E3", N, c5.h, c5.w, 3 * hid)
            self.conv("aspp.pw", c5, [b.conv[0][0] for b in branches], [b.conv[0][1] for b in branches], e3, R6)
            if self.prec_name == "f32" and int(os.environ.get("UAVSAL_ASPP_GROUP", "1")) and hid % 32 == 0:
                # ... and their three projections (1920 -> 256 each, different inputs) are ONE launch too: output-channel
                # groups with their own A columns (uavsal_conv_desc.n_group), K shared out over workgroups
                d3 = self._scr("D3", N, c5.h, c5.w, 3 * hid)
                if ASPP_DW_MERGE and all(b.stride == 1 for b in branches) and hid % 64 == 0:
                    # ... and so are their three dilated depthwise convs: channel groups with their own dilation in the whole-map
                    # kernel (uavsal_dw_desc.dil_group_c).  Three launches on three lanes cost six event operations on the main
                    # stream (~25 us between aspp.pw and aspp.pl) for ~10 us of overlap
                    self.dw("aspp.dw", e3, [b.conv[1][0] for b in branches], [b.conv[1][1] for b in branches], d3, 1,
                            [getattr(b, "dilation", 1) for b in branches])
                    aspp_dw_merged = True
                else:
                    for bi, b in enumerate(branches):
                        fork(3 + bi)
                        self.dw("aspp%d.dw" % (bi + 2), e3.slice(bi * hid, hid), b.conv[1][0], b.conv[1][1], d3.slice(bi * hid, hid),
                                b.stride, getattr(b, "dilation", 1))
                        self.main()
                aspp_grouped = d3
            else:
                aspp_grouped = None
                for bi, b in enumerate(branches):
                    fork(3 + bi)
                    self.ir_block("aspp%d" % (bi + 2), c5, b, aspp.slice(256 * (bi + 1), 256), expanded=e3.slice(bi * hid, hid))
                    self.main()
        else:
            aspp_grouped = None
            for bi, b in enumerate(branches):
                fork(3 + bi)
                self.ir_block("aspp%d" % (bi + 2), c5, b, aspp.slice(256 * (bi + 1), 256))
                self.main()
        if seg_mode != 2:
            fork(6)
        self.conv("conv_lv4", c4, sf.conv_lv4[0], sf.conv_lv4[1], x4, R6)
        if not seg_mode:
            self.bilinear("up_c4", x4, cat.slice(256, 128))
        self.conv("conv_lv3", c3, sf.conv_lv3[0], sf.conv_lv3[1], lv3, R6)
        if seg_mode != 2:
            self.main()
        self.conv("aspp1", c5, sf.lv5_aspp1[0], sf.lv5_aspp1[1], aspp.slice(0, 256), R6)
        if not aspp_dw_merged:
            join(3)
            join(4)
            join(5)
        if aspp_grouped is not None:
            hid = branches[0].hidden
            self.conv("aspp.pl", aspp_grouped.slice(0, hid), [b.conv[2] for b in branches], [b.conv[3] for b in branches],
                      aspp.slice(256, 768), NONE, cout=768, n_group=256)
        self.conv("conv_lv5", aspp, sf.conv_lv5[0], sf.conv_lv5[1], x5, R6)
        if not seg_mode:
            self.bilinear("up_c5", x5, cat.slice(0, 256))
        if seg_mode != 2:
            join(6)
        x = self._buf("sfnet", N, h, w, 256)
        if seg_mode:
            self.conv3_wino("conv_last", V(None, N, h, w, 448), sf.conv_last[0], sf.conv_last[1], x, R6, r=self.winograd_r,
                            segs=[x5, x4, lv3])
        elif wino_last:
            self.conv3_wino("conv_last", cat, sf.conv_last[0], sf.conv_last[1], x, R6, r=self.winograd_r)
        else:
            self.conv("conv_last", cat, sf.conv_last[0], sf.conv_last[1], x, R6, taps=9)
        self._mark("srf_head", s0)

        # ---- ST blocks (model.py:235-249)
        s0 = len(self.ops_meta)
        for i, st in enumerate(m.st_layer):
            sp = self._buf("st%d_sp" % i, N, h, w, 256)
            te = st.stconv_te
            r = self._buf("st%d_red" % i, N, h, w, 32)
            dif = self._buf("st%d_dif" % i, N, h, w, 64)
            t1 = self._buf("st%d_te1" % i, N, h, w, 32)
            # temporal branch (small launches): st_lanes 1 = all of it on lane 6, next to the spatial branch's big GEMMs;
            # 2 = its first two launches on the main lane (they would otherwise queue behind a grid-filling GEMM for
            # the whole of it), the rest on lane 6; 0 = no side lane
            # (round 2, same box, two runs each: 5.26 / 5.25 / 5.22 ms for 1 / 0 / 2.  Round 5, one clip: 4.234 / 4.221 for 2 / 0 --
            # the fork / join pair costs more than the overlap buys while the spatial branch's GEMMs fill the chip anyway; eight
            # clips: 27.97-28.10 / 28.31 for 2 / 0.  A function of the frame count only)
            st_lanes = int(os.environ.get("UAVSAL_ST_LANES", "0" if N <= 8 else "2"))
            if st_lanes == 1:
                self.fork(6)
            self.conv("st%d.reduce" % i, x, te.reduce_conv[0], te.reduce_conv[1], r, R6)
            self._meta(kind="tdiff", name="st%d.tdiff" % i, flops=0.0, bytes=4.0 * N * hw * 96)
            self._touch(r, dif)
            if not self._dry:
                d = L.TdiffDesc()
                d.inp, d.ldi, d.out, d.ldo = r.ptr, 32, dif.ptr, 64
                d.n_img, d.HW, d.C, d.seq_len = N, hw, 32, self.seq_len
                self._add(self.lib.uavsal_plan_add_tdiff, d, "plan_add_tdiff")
            if st_lanes == 2:
                self.fork(6)
            self.ir_block("st%d.sub" % i, dif, te.sub_conv, t1)
            if st_lanes:
                self.main()
            self.ir_block("st%d.sp" % i, x, st.stconv_sp.spconv, sp)
            if st_lanes:
                self.join(6)
            ssum = self._buf("st%d_sum" % i, N, h, w, 256)
            self.conv("st%d.te_last" % i, t1, te.last_conv[0], te.last_conv[1], ssum, R6, res=sp)   # x_sp + x_te
            y = self._buf("st%d" % i, N, h, w, 256)
            self.conv("st%d.last" % i, ssum, st.stconv_last[0], st.stconv_last[1], y, R6, res=x)    # x + out
            x = y
        self._mark("st_blocks", s0)

        # ---- fuse + multi-prior net (model.py:344-365)
        s0 = len(self.ops_meta)
        if not num_cb:                   # no prior at all: the recurrence reads fust_layer's output (model.py:346, 367)
            xf = self._buf("prefuse", N, h, w, 256)
            self.ir_block("fust", x, m.fust_layer[0], xf)
        else:
            fu = self._buf("fu320", N, h, w, 320)
            xs = fu.slice(0, 256)
            self.ir_block("fust", x, m.fust_layer[0], xs)
            if use_c:
                B = N // self.ctx_T
                tsum = self._buf("ctx_sum", B, h, w, 256)
                self._meta(kind="tsum", name="ctx.sum", flops=0.0, bytes=4.0 * (N + B) * hw * 256)
                self._touch(xs, tsum)
                if not self._dry:
                    d = L.TsumDesc()
                    d.inp, d.ldi, d.out, d.ldo = xs.ptr, 320, tsum.ptr, 256
                    d.n_groups, d.T, d.HW, d.C = B, self.ctx_T, hw, 256
                    self._add(self.lib.uavsal_plan_add_tsum, d, "plan_add_tsum")
                h2, w2 = _down(h), _down(w)
                cx1 = self._buf("ctx1", B, h2, w2, 64)
                self.ir_block("ctx.0", tsum, m.cxt_cb_prior[0], cx1)
                h3, w3 = _down(h2), _down(w2)
                cx2 = self._buf("ctx2", B, h3, w3, 64)
                self.ir_block("ctx.1", cx1, m.cxt_cb_prior[1], cx2)
                cslot = cb.slice(cb_off["ctx"], 64)
                if self.ctx_mode == "tile":      # cb_cxt.repeat(T,1,1,1): frame k <- chunk k % B (model.py:361)
                    self.bilinear("ctx.up", cx2, cslot, src_mod=B, src_div=1)
                else:                            # independent clips: frame (c,t) <- clip c
                    self.bilinear("ctx.up", cx2, cslot, src_mod=N, src_div=self.ctx_T)
            for lane in self._prior_lanes:
                self.join(lane)
            self.ir_block("fucb", cb, m.fucb_layer[0], fu.slice(256, 64))
            self.named["fust_in_cb"] = fu.slice(256, 64)
            xf = self._buf("prefuse", N, h, w, 256)
            self.ir_block("fucbst", fu, m.fucbst_layer[0], xf)
        self._mark("prior_fuse", s0)

        # ---- recurrence: ConvTWA (model_convlstm.py:276-292, 368-371) or ConvLSTM (:111-126, 206-222)
        s0 = len(self.ops_meta)
        rc = m.rnn.cell_list[0].rnn_conv
        Lq = self.seq_len
        ro = self._buf("rnn", N, h, w, 256)
        lstm = getattr(m, "rnn_type", "twa") == "lstm"
        if lstm:
            pre = self._buf("lstm_pre", N, h, w, 1024)      # W[:, :256] * x_t for all t, rows 4*c+gate
            self.conv("lstm.wx", xf, rc, None, pre, NONE, taps=9, wslice=(0, 256), gate_interleave=256)
            co = self._buf("lstm_c", N, h, w, 256)          # cell-state history
            for t in range(Lq):
                hp = self.named["h0"] if t == 0 else ro.frames(t - 1, self.n_seq)
                cp = c0 if t == 0 else co.frames(t - 1, self.n_seq)
                a = V(hp.t, self.n_seq, h, w, 256, 256, hp.coff)
                st = hw if t == 0 else Lq * hw
                strides = {"a": st, "r": st, "o": Lq * hw, "x": Lq * hw}
                self.conv("lstm.step%d" % t, a, rc, None, ro.frames(t, self.n_seq), NONE, taps=9,
                          wslice=(256, 512), gate_interleave=256, epi=L.EPI_LSTM, cout=1024,
                          res=V(cp.t, self.n_seq, h, w, 256, 256, cp.coff), aux=pre.frames(t, self.n_seq),
                          out2=co.frames(t, self.n_seq), n_img=self.n_seq, strides=strides)
            self.named["lstm_c"] = co
        else:
            pre = self._buf("twa_pre", N, h, w, 256)
            if self.winograd and self._prec_for("twa.wx") == "f32":
                self.conv3_wino("twa.wx", xf, rc, None, pre, NONE, wslice=(0, 256), r=self.winograd_r)
            else:
                self.conv("twa.wx", xf, rc, None, pre, NONE, taps=9, wslice=(0, 256))    # W[:, :256] * x_t, all t
        for t in range(0 if lstm else Lq):
            a = h0 if t == 0 else ro.frames(t - 1, self.n_seq)
            a = V(a.t, self.n_seq, h, w, 256, 256, a.coff)
            strides = {"a": hw if t == 0 else Lq * hw, "o": Lq * hw, "r": Lq * hw, "x": Lq * hw}
            # (split-fp16 plans from four clips up: the per-step gate convolution in exact fp32 through Winograd F(4x4) as well -- 1.78x
            # fewer MFMA FLOPs than the direct 3x3 and it beats the split-fp16 implicit GEMM there: 136 vs 160 us per step at eight
            # clips, 17.89 -> 17.67 ms per eight-clip step; more accurate, never less.  F16X3_WINO_STEPS = 0: the direct split-fp16 step)
            f16_wino = (F16X3_WINO_STEPS and self.prec_name == "f16x3" and not self.prec_overrides and self.n_seq >= 4
                        and bool(getattr(m, "winograd", True)))
            if (self.winograd or f16_wino) and self.winograd_steps and (f16_wino or self._prec_for("twa.step") == "f32"):
                # one clip: 920 tiles of 2x2 fill the chip with 128x128 GEMM tiles; four clips and more: F(4x4) (1.78x
                # fewer FLOPs, smaller transforms) on 64x64 tiles (measured: 4.54 vs 4.61 ms at one clip, 29.47 vs 28.80 at eight)
                many = self.n_seq >= 4
                self.conv3_wino("twa.step%d" % t, a, rc, None, ro.frames(t, self.n_seq), NONE, wslice=(256, 512),
                                n_img=self.n_seq, strides=strides, twa=(xf.frames(t, self.n_seq), pre.frames(t, self.n_seq)),
                                gemm_tile=self.winograd_steps if self.winograd_steps > 0 else (11 if many else 8),
                                r=self.winograd_step_r or (4 if many else 2))
                continue
            self.conv("twa.step%d" % t, a, rc, None, ro.frames(t, self.n_seq), NONE, taps=9, wslice=(256, 512),
                      epi=L.EPI_TWA, res=xf.frames(t, self.n_seq), aux=pre.frames(t, self.n_seq),
                      n_img=self.n_seq, strides=strides)
        self._mark("twa", s0)

        # ---- decoder + sigmoid (model.py:372-373), state back to NCHW
        s0 = len(self.ops_meta)
        outv = V(_Fake() if self._dry else self.out, N, h, w, 1)
        if self.keep_taps:
            lv = V(_Fake() if self._dry else self.logits, N, h, w, 1)
            self.ir_block("conv_out_st.logits", ro, m.conv_out_st, lv, final_act=NONE)
        self.ir_block("conv_out_st", ro, m.conv_out_st, outv, final_act=L.ACT_SIGMOID)
        outs = [(ro, "state_out", h0)] + ([(self.named["lstm_c"], "cstate_out", c0)] if lstm else [])
        if self.persistent:
            # h_last of every clip (NHWC rows of the history) -> the resident state buffer, one strided copy
            for hist, _, keep in outs:
                self._meta(kind="copy", name="state.keep", flops=0.0, bytes=8.0 * self.n_seq * 256 * hw)
                self._touch(hist, keep)
                if not self._dry:
                    d = L.CopyDesc()
                    d.inp, d.out = hist.frames(Lq - 1, 1).ptr, keep.ptr
                    d.in_pitch, d.out_pitch, d.row_floats, d.rows = Lq * hw * 256, hw * 256, hw * 256, self.n_seq
                    self._add(self.lib.uavsal_plan_add_copy, d, "plan_add_copy")
        else:
            for c in range(self.n_seq):
                for hist, dst, _ in outs:
                    last = hist.frames(c * Lq + Lq - 1, 1)
                    nm = "%s%d" % (dst.replace("_", "."), c)           # state.out0, cstate.out0, ...
                    self.layout(nm, last, None if self._dry else getattr(self, dst).data_ptr() + 4 * c * 256 * hw, 1, 256, hw, 256, 0)
        # error guard: poisons what the caller will see if any kernel of this run set the error word
        self._meta(kind="guard", name="guard", flops=0.0, bytes=0.0)
        if self.persistent:
            self._touch(h0, c0)
        if not self._dry:
            bufs = [(self.out.data_ptr(), self.out.numel())]
            if self.persistent:
                bufs.append((h0.ptr, self.n_seq * hw * 256))
                bufs.append((c0.ptr, self.n_seq * hw * 256) if lstm else (None, 0))
                # what the caller gets back: channels-last views of the resident buffers
                self.h_view = h0.t.view(self.n_seq, h, w, 256).permute(0, 3, 1, 2)
                self.c_view = c0.t.view(self.n_seq, h, w, 256).permute(0, 3, 1, 2) if lstm else None
            else:
                bufs.append((self.state_out.data_ptr(), self.state_out.numel()))
                bufs.append((self.cstate_out.data_ptr(), self.cstate_out.numel()) if lstm else (None, 0))
            r = self.lib.uavsal_plan_add_guard(self.plan, bufs[0][0], bufs[0][1], bufs[1][0], bufs[1][1], bufs[2][0], bufs[2][1])
            if r < 0:
                L.check(r, "plan_add_guard")
        self._mark("decoder", s0)

    # ------------------------------------------------------------------ execution
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def launch(self):
        """Launch the recorded plan once on torch's current stream (no staging, no sync)."""
        if self.use_graph:
            # the legacy default stream cannot be captured: capture and replay on a private
            # stream, fenced against torch's current stream on both sides
            cur = torch.cuda.current_stream(self.device)
            if not self._graph_ready:
                self._gstream = torch.cuda.Stream(self.device)
                self._gstream.wait_stream(cur)
                gs = C.c_void_p(self._gstream.cuda_stream)
                L.check(self.lib.uavsal_plan_run(self.plan, 0, -1, gs), "plan_run (warm-up)")
                self._gstream.synchronize()
                L.check(self.lib.uavsal_plan_graph_build(self.plan, gs), "plan_graph_build")
                self._graph_ready = True
            self._gstream.wait_stream(cur)
            L.check(self.lib.uavsal_plan_graph_launch(self.plan, C.c_void_p(self._gstream.cuda_stream)),
                    "plan_graph_launch")
            cur.wait_stream(self._gstream)
        else:
            if self.inplace and not self._bound:
                raise RuntimeError("launch-loop plan was never bound to the caller's tensors (Engine.run does that)")
            L.check(self.lib.uavsal_plan_run(self.plan, 0, -1, self._stream()), "plan_run")

    def _is_resident(self, t, view) -> bool:
        return (t is not None and view is not None and t.data_ptr() == view.data_ptr()
                and t.numel() == view.numel() and t.reshape(view.shape).stride() == view.stride())

    def _stage_state_persistent(self, state, cstate):
        """Persistent mode: the caller's state is already resident when it is the view `run` returned (or a
        detach of it, Demo_Test.py:86); None resets it (model_convlstm.py:356); any other tensor is loaded."""
        pairs = [(state, self.h_view, self.state_in, "h0")]
        if self.c_view is not None:
            pairs.append((cstate, self.c_view, self.cstate_in, "c0"))
        for t, view, stage, name in pairs:
            if self._is_resident(t, view):
                continue
            buf = self.named[name]
            if t is None:
                buf.t.zero_()
                continue
            stage.copy_(t.reshape(stage.shape))
            d = L.LayoutDesc()
            d.inp, d.out, d.n_img, d.C, d.HW, d.ld, d.to_nhwc, d.Cpad = (
                stage.data_ptr(), buf.ptr, self.n_seq, 256, self.h * self.w, 256, 1, 0)
            L.check(self.lib.uavsal_layout(C.byref(d), self._stream()), "uavsal_layout(state)")

    def _bind_in_place(self, x, cb0, cb1, state, cstate, lstm):
        """Point the plan at the caller's tensors and at freshly allocated outputs (no staging copies, no
        clones).  Non-contiguous inputs are made contiguous first; everything bound is kept referenced until
        the next call."""
        dev = self.device
        x = x.reshape(self.x_in.shape).contiguous()
        hold = [x]
        self._patch("features.0", 1 if self.in_dtype == torch.uint8 else 0, x.data_ptr())
        for t, stage, op, on in ((cb0, self.cb0_in, "gauss.in", self.use_priors[0]), (cb1, self.cb1_in, "ob.in", self.use_priors[1])):
            if not on:                  # a prior this model does not have is never read (reference model.py:347-353)
                continue
            if self.static_priors:      # (zero frame stride, checked by the model: frame 0 is every frame)
                t = t[:1]
            t = t.reshape(stage.shape).contiguous()
            self._patch(op, 0, t.data_ptr())
            hold.append(t)
        out = torch.empty((self.N, self.h * self.w), dtype=torch.float32, device=dev)
        # (the decoder's last launch: ".dwpl" when its depthwise runs inside the projection)
        self._patch("conv_out_st.pl" if "conv_out_st.pl" in self._op_idx else "conv_out_st.dwpl", 1, out.data_ptr())
        self._patch("guard", 0, out.data_ptr())
        hold.append(out)
        st = None
        if self.persistent:
            self._stage_state_persistent(state, cstate)
        else:
            shape = self.state_in.shape
            per = 4 * 256 * self.h * self.w
            pairs = [("state", state, 1)] + ([("cstate", cstate, 2)] if lstm else [])
            outs = []
            for nm, t, gslot in pairs:
                t = self.zero_state if t is None else t.reshape(shape).contiguous()
                self._patch(nm + ".in", 0, t.data_ptr())
                o = torch.empty(shape, dtype=torch.float32, device=dev)
                for c in range(self.n_seq):
                    self._patch("%s.out%d" % (nm, c), 1, o.data_ptr() + c * per)
                self._patch("guard", gslot, o.data_ptr())
                hold += [t, o]
                outs.append(o)
            st = (outs[0], outs[1]) if lstm else outs[0]
        self._hold = hold
        self._bound = True
        return out, st

    def stage_inputs(self, x, cb0, cb1, state, cstate=None):
        self.x_in.copy_(x.reshape(self.x_in.shape))
        for t, stage, on in ((cb0, self.cb0_in, self.use_priors[0]), (cb1, self.cb1_in, self.use_priors[1])):
            if on:
                stage.copy_((t[:1] if self.static_priors else t).reshape(stage.shape))
        if self.persistent:
            return self._stage_state_persistent(state, cstate)
        if state is None:
            self.state_in.zero_()
        else:
            self.state_in.copy_(state.reshape(self.state_in.shape))
        if cstate is None:
            self.cstate_in.zero_()
        else:
            self.cstate_in.copy_(cstate.reshape(self.cstate_in.shape))

    def run(self, x, cb0, cb1, state=None, taps: Optional[dict] = None, cstate=None):
        if x.dtype != self.in_dtype:
            raise RuntimeError("engine built for %s frames, got %s" % (self.in_dtype, x.dtype))
        lstm = getattr(self.model, "rnn_type", "twa") == "lstm"
        with torch.cuda.device(self.device):
            self.check(wait=False)               # a previous asynchronous run that is over by now
            if self.inplace:
                out, st = self._bind_in_place(x, cb0, cb1, state, cstate, lstm)
            else:
                self.stage_inputs(x, cb0, cb1, state, cstate)
            self.launch()
            if self.sync_errors:
                self.check(wait=True)
            if not self._first_run_verified and self.split_mode:
                # once per plan (this first call is therefore synchronous even with sync_errors off): a split shadow
                # nobody wrote (see _buf) shows up as NaN in the maps.  Device errors are raised first, under their own
                # name: the guard's NaN fill after a stream-K time-out must not be reported as a missing shadow
                self._first_run_verified = True
                self.check(wait=True)
                res = out if self.inplace else self.out
                if bool(torch.isnan(res).any().item()) and not bool(torch.isnan(x.float()).any().item()):
                    raise RuntimeError("the first run of this plan produced NaN maps from finite frames: a split shadow "
                                       "was read that no producer wrote (engine._no_shadow is missing a buffer)")
            if not self.inplace:
                out = self.out.clone()
                if self.persistent:
                    st = None
                else:
                    st = self.state_out.clone()
                    if lstm:
                        st = (st, self.cstate_out.clone())
            if self.persistent:          # opt-in aliasing: views of the resident state, overwritten by the next call
                st = (self.h_view, self.c_view) if lstm else self.h_view
            if taps is not None:
                if not self.keep_taps:
                    raise RuntimeError("engine was built without taps")
                for k in ("c3", "c4", "c5", "sfnet", "st0", "st1", "fust_in_cb", "prefuse", "rnn"):
                    if k in self.named:          # (no "fust_in_cb" in a model without priors)
                        taps[k] = self.tap(k)
                taps["logits"] = self.logits.clone().view(self.N, 1, self.h, self.w)
        return out, st

    def state_split(self) -> int:
        """Index of the first op that reads the recurrent state (the recurrence's first step; in resident-state mode nothing in
        front of it does): ops [0, state_split) of a forward do not depend on the previous forward of the same video."""
        for i, m in enumerate(self.ops_meta):
            if m.get("name", "").startswith(("twa.step0", "lstm.step0")):
                return i
        raise RuntimeError("plan has no recurrence step")

    def run_streamed(self, x, cb0, cb1, prev: Optional["Engine"] = None, prev_done: Optional[torch.cuda.Event] = None, reset=False):
        """One group of a VIDEO whose previous group ran (or is still running) on `prev` -- this engine or another replica's --
        in resident-state mode: everything in front of the recurrence is launched at once, then the launch stream waits for
        `prev_done`, takes over `prev`'s recurrent state (a device copy of the NHWC state buffer; `reset`: zeros -- the first
        group, model_convlstm.py:356) and runs the recurrence, the decoder and the guard.  The arithmetic of a group is that of
        `run` (same plan, same order per lane-less launch): maps are bit-identical to the sequential loop; what changes is that
        the head of group k + 1 overlaps the tail of group k (Demo_Test.py:75-86 runs them back to back)."""
        if not (self.persistent and self.inplace):
            raise RuntimeError("run_streamed needs the resident-state launch-loop plan (model.persistent_state = True, no graph)")
        lstm = getattr(self.model, "rnn_type", "twa") == "lstm"
        with torch.cuda.device(self.device):
            self.check(wait=False)
            out, _ = self._bind_in_place(x, cb0, cb1, self.h_view, self.c_view, lstm)      # (resident views: nothing is staged)
            split = self.state_split()
            L.check(self.lib.uavsal_plan_run(self.plan, 0, split, self._stream()), "plan_run(head)")
            cur = torch.cuda.current_stream(self.device)
            if prev_done is not None:
                cur.wait_event(prev_done)
            for name in ("h0", "c0") if lstm else ("h0",):
                mine = self.named[name].t
                if reset:
                    mine.zero_()
                elif prev is not None and prev is not self:
                    mine.copy_(prev.named[name].t)
            L.check(self.lib.uavsal_plan_run(self.plan, split, -1, self._stream()), "plan_run(tail)")
        return out

    def check(self, wait=True):
        """Raise if the most recent run reported a device-side error (its outputs were overwritten with NaN).
        `wait=False` only looks when that run is known to have finished."""
        code = self.lib.uavsal_plan_status(self.plan, 1 if wait else 0)
        if code == 0:
            return
        if code == -5:
            # a piece published after its owner gave up would be consumed by the next launch: start clean
            torch.cuda.synchronize(self.device)
            for ws in self._sk_ws.values():
                ws[:65536].zero_()
            raise RuntimeError("UAVSal HIP path: a stream-K hand-off timed out on the device "
                               "(UAVSAL_ERR_STREAMK); the maps and state of that call are invalid (NaN-filled)")
        L.check(code, "uavsal_plan_status")

    def streamk_clean(self) -> bool:
        """True when the stream-K workspace's flag block is all zero, as every launch must leave it (a set
        word is a published piece that nobody collected, or a bounded wait that gave up).  Synchronises."""
        torch.cuda.synchronize(self.device)
        return all(int(ws[:65536].view(torch.int32).abs().sum().item()) == 0 for ws in self._sk_ws.values())

    def tap(self, name) -> torch.Tensor:
        """NCHW copy of a named NHWC buffer (debug / parity tests)."""
        v = self.named[name]
        base = v.t.tensor() if isinstance(v.t, _ArenaRef) else v.t
        t = base.view(v.n, v.h, v.w, v.ld) if base.numel() == v.n * v.h * v.w * v.ld else None
        if t is None:
            raise RuntimeError("tap %s is not a dense buffer" % name)
        return t[..., v.coff:v.coff + v.c].permute(0, 3, 1, 2).contiguous()

    def run_ops(self, first, last):
        """Launch ops [first, last) once, flat on the current stream (no lanes, no staging): puts the activations an op reads
        back in place before it is timed by itself -- buffers share addresses by liveness."""
        if last > first:
            L.check(self.lib.uavsal_plan_run(self.plan, first, last, self._stream()), "plan_run")

    def time_ops(self, first, last, iters=10) -> float:
        """Average device milliseconds for ops [first,last) measured with hipEvents on the launch stream."""
        ms = C.c_float(0.0)
        L.check(self.lib.uavsal_plan_time(self.plan, first, last, iters, self._stream(), C.byref(ms)), "plan_time")
        return float(ms.value)


class _Fake:
    """Stand-in tensor for the sizing pass."""
    def data_ptr(self):
        return 0

    def numel(self):
        return 0

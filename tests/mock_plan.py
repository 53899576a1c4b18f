"""Record a launch plan WITHOUT a device: the library's shape queries are the real ones (they need no GPU), the plan-recording
entry points are stubs that count the ops and keep the fill descriptors.  Lets the CPU suite run the engine's second pass --
where every activation address is handed out by the arena and checked against the buffer's declared live range."""
import torch

from iip_uavsal_saliency_amd import _lib as L
from iip_uavsal_saliency_amd import engine as E


class MockLib:
    def __init__(self, real):
        self.real, self.n, self.fills, self.lanes = real, 0, [], []
        self.cur_lane = 0

    def __getattr__(self, name):
        if name.startswith("uavsal_plan_add_"):
            def add(plan, *a):
                if name == "uavsal_plan_add_fill":
                    d = a[0]._obj
                    self.fills.append((self.n, int(d.out), int(d.n), self.cur_lane))
                self.n += 1
                return self.n - 1
            return add
        if name == "uavsal_plan_set_lane":
            def set_lane(plan, lane):
                self.cur_lane = lane
                return 0
            return set_lane
        if name == "uavsal_plan_create":
            return lambda: 1
        if name == "uavsal_plan_error_word":
            return lambda p: 4096
        if name in ("uavsal_plan_enable_lanes", "uavsal_plan_destroy", "uavsal_plan_patch_ptr"):
            return lambda *a: 0
        return getattr(self.real, name)


class _NoDevice:
    def __init__(self, d):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def record(model, **kw):
    """(engine, mock library) of a plan recorded on the CPU (activations in host memory, nothing launched)."""
    mock = MockLib(L.load())
    orig_load, orig_dev = L.load, torch.cuda.device
    L.load, torch.cuda.device = (lambda: mock), _NoDevice
    try:
        eng = E.Engine(model, "cpu", plan_only=True, **kw)       # sizing pass + placement ...
        eng.plan_only = False                                   # ... then the recording pass against the stubs
        eng._init_on_device(True, resume=True)
    finally:
        L.load, torch.cuda.device = orig_load, orig_dev
    return eng, mock

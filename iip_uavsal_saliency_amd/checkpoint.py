"""Loading the reference's published checkpoints without the reference on the path.

The reference saves WHOLE pickled models (`torch.save(model, path)`, Demo_Train_Test.py:159-160,174)
and loads them with `torch.load(model_path).state_dict()` (Demo_Test.py:39, model.py:339).
Unpickling such a file normally needs importable classes `model.UAVSal`, `model.dwBlock`,
`model_convlstm.ConvTWA`, `model_feature.ReMobileNetV2` and torchvision's
`ConvBNReLU` / `InvertedResidual`.  `load_reference_state_dict` swaps in
an unpickler that resolves every class from those modules to an inert `nn.Module` stand-in (a
pickled module is just its `__dict__`), then returns `state_dict()` -- keys and shapes are the
reference's, so the result feeds `UAVSal.load_state_dict` directly.  torch's own classes
(`torch.nn.Conv2d`, tensors, storages) resolve normally.
"""
from __future__ import annotations

import pickle
from typing import Dict

import torch
import torch.nn as nn

_SHIMMED_PREFIXES = ("model", "model_feature", "model_convlstm", "torchvision")


class _ShimUnpickler(pickle.Unpickler):
    _cache: Dict[tuple, type] = {}

    def find_class(self, module, name):
        root = module.split(".")[0]
        if root in _SHIMMED_PREFIXES:
            key = (module, name)
            if key not in self._cache:
                self._cache[key] = type(name, (nn.Module,), {"__module__": "uavsal_ckpt_shim." + module,
                                                             "forward": lambda self, *a, **k: None})
            return self._cache[key]
        return super().find_class(module, name)


class _ShimPickleModule:
    """Duck-typed `pickle_module` for torch.load."""
    Unpickler = _ShimUnpickler
    load = staticmethod(lambda f, **kw: _ShimUnpickler(f, **kw).load())
    __name__ = "uavsal_ckpt_shim"


def load_reference_state_dict(path: str) -> Dict[str, torch.Tensor]:
    """`path`: a reference checkpoint (whole pickled `UAVSal`) or a plain state_dict file."""
    obj = torch.load(path, map_location="cpu", pickle_module=_ShimPickleModule, weights_only=False)
    if isinstance(obj, dict):
        return obj
    if not isinstance(obj, nn.Module):
        raise ValueError("unexpected checkpoint content: %r" % type(obj))
    return obj.state_dict()


def load_reference_checkpoint(model, path: str, strict: bool = True):
    """`model.load_state_dict(torch.load(path).state_dict())` of Demo_Test.py:39 for the drop-in model."""
    return model.load_state_dict(load_reference_state_dict(path), strict=strict)

"""Build libuavsal_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
INCLUDE = os.path.join(ROOT, "include")
LIB = os.path.join(PKG, "libuavsal_hip.so")
SOURCES = ["conv_gemm.hip", "dw_conv.hip", "fused_ir.hip", "glue.hip", "post.hip", "plan.hip"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-I" + INCLUDE, "-I" + CSRC]
FLAGS += os.environ.get("UAVSAL_EXTRA_HIPCC_FLAGS", "").split()        # e.g. -DUAVSAL_PROBE for tools/gemm_probe2.py


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 and link the shared library.  Returns its path."""
    headers = [os.path.join(CSRC, "common.h"), os.path.join(INCLUDE, "uavsal_hip.h")]
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        if force or not _newer(o, [s] + headers):
            jobs.append([_hipcc()] + FLAGS + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr[-4000:]))

    if jobs:
        with ThreadPoolExecutor(max_workers=4) as ex:
            list(ex.map(run, jobs))
    if jobs or force or not _newer(LIB, objs):
        run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

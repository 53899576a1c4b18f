"""Build libuavsal_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
INCLUDE = os.path.join(ROOT, "include")
LIB = os.path.join(PKG, "libuavsal_hip.so")
SOURCES = ["conv_gemm.hip", "conv_gemm_k32.hip", "dwproj.hip", "dw_conv.hip", "fused_ir.hip", "fused_mid.hip", "glue.hip", "post.hip", "plan.hip", "winograd.hip"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-I" + INCLUDE, "-I" + CSRC]
FLAGS += os.environ.get("UAVSAL_EXTRA_HIPCC_FLAGS", "").split()        # e.g. -DUAVSAL_PROBE for tools/gemm_probe2.py


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _digest(paths, extra="") -> str:
    h = hashlib.sha256(extra.encode())
    for p in paths:
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _stamp_ok(stamp: str, digest: str, target: str) -> bool:
    try:
        return os.path.exists(target) and open(stamp).read().strip() == digest
    except OSError:
        return False


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source for gfx950 and link the shared library.  Returns its path.
    An object is reused only when the CONTENT of its source, of every header under csrc/ and include/, and the
    compiler flags are what they were when it was built (a `.stamp` beside it holds their hash): file times are not
    trusted, the objects travel to the GPU box outside git."""
    headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    headers += sorted(os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h"))
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        digest = _digest([s] + headers, " ".join(FLAGS))
        if force or not _stamp_ok(o + ".stamp", digest, o):
            jobs.append(([_hipcc()] + FLAGS + ["-c", s, "-o", o], o + ".stamp", digest))

    def run(job):
        cmd, stamp, digest = job
        if verbose:
            print(" ".join(cmd), flush=True)
        if stamp and os.path.exists(stamp):
            os.remove(stamp)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr[-4000:]))
        if stamp:
            with open(stamp, "w") as f:
                f.write(digest)

    if jobs:
        with ThreadPoolExecutor(max_workers=4) as ex:
            list(ex.map(run, jobs))
    link_digest = _digest(objs, "link")
    if jobs or force or not _stamp_ok(LIB + ".stamp", link_digest, LIB):
        run(([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs, LIB + ".stamp", link_digest))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

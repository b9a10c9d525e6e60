"""Build libmoyolo.so (hipcc, gfx950 only) in-tree so it travels with the repo snapshot."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmoyolo.so")
SOURCES = ["gemm.hip", "gemm_wreg.hip", "gemm_dma.hip", "conv_ws.hip", "stem_l1.hip", "c2f_fused.hip", "mlp_head.hip", "dec_tail.hip", "dec_mid.hip", "msda_raw.hip", "ops.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libmoyolo.so cannot be built (ROCm toolchain required)")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in _deps())


def _deps():
    """Sources only: .hip/.hpp under csrc plus the public header (never objects or compiler temporaries)."""
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".hpp", ".h"))]
    return srcs + [os.path.join(os.path.dirname(HERE), "include", "moyolo.h")]


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = _hipcc()
    objs, procs = [], []
    hdr_t = max(os.path.getmtime(d) for d in _deps() if not d.endswith(".hip"))
    for src in SOURCES:
        path = os.path.join(CSRC, src)
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), hdr_t):
            continue
        cmd = [hipcc, *FLAGS, "-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))      # the translation units are independent: compile them side by side
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)

"""Build the HIP libraries (hipcc, gfx950 only) in-tree so they travel with the repo snapshot.

    python -m mo_yolo_amd.build            # libmoyolo.so       -- the PRODUCT: reads no environment variable, holds no timing-only kernel
    python -m mo_yolo_amd.build --diag     # libmoyolo_diag.so  -- the LAB: same sources with -DMOY_DIAG=1 (A/B knobs `MOY_*`, stamped /
                                           #                       ablated template instances); selected with MOYOLO_LIB=<path>
    python -m mo_yolo_amd.build --all [--force]

Every translation unit is compiled with -Rpass-analysis=kernel-resource-usage and the remarks are checked: a kernel of an
LDS-DMA ring family (`gemm_wreg_kernel`, `conv_ws_kernel`, `conv_s2_kernel`, `gemm_dma_kernel`, ...) that needs scratch FAILS the build --
its hand-counted `s_waitcnt vmcnt(N)` bookkeeping knows nothing of scratch loads / stores (the round-4 race was found in the shadow of
exactly such forms); scratch in any other kernel is listed in `csrc/obj*/resources.txt` beside the register counts.
"""
from __future__ import annotations

import os
import re
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmoyolo.so")
LIB_DIAG = os.path.join(HERE, "libmoyolo_diag.so")
SOURCES = ["gemm.hip", "gemm_wreg.hip", "gemm_dma.hip", "conv_ws.hip", "stem_l1.hip", "c2f_fused.hip", "mlp_head.hip", "dec_tail.hip", "dec_mid.hip", "msda_raw.hip", "ops.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-Wall", "-Wno-unused-function", "-Rpass-analysis=kernel-resource-usage"]
# kernels whose vector-memory queue is bookkept by hand (counted vmcnt): no scratch traffic allowed
RING_KERNELS = ("gemm_wreg_kernel", "conv_ws_kernel", "conv_s2_kernel", "conv_ws_pp_kernel", "conv_ws_pipe_kernel", "gemm_dma_kernel")


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libmoyolo.so cannot be built (ROCm toolchain required)")


def lib_path(diag=False):
    return LIB_DIAG if diag else LIB


def _objdir(diag):
    return os.path.join(CSRC, "obj_diag" if diag else "obj")


def needs_build(diag=False):
    lib = lib_path(diag)
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(d) > t for d in _deps())


def _deps():
    """Sources only: .hip/.hpp under csrc plus the public header (never objects or compiler temporaries)."""
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".hpp", ".h"))]
    return srcs + [os.path.join(os.path.dirname(HERE), "include", "moyolo.h"), os.path.abspath(__file__)]


_REMARK = re.compile(r"remark: (?:Function Name: (?P<fn>\S+)|\s+(?P<key>VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (?P<val>\d+))")


def kernel_resources(remarks: str):
    """[(mangled kernel name, {VGPRs, AGPRs, ScratchSize, Occupancy, LDS})] from the -Rpass-analysis=kernel-resource-usage remarks."""
    out, cur = [], None
    for line in remarks.splitlines():
        m = _REMARK.search(line)
        if not m:
            continue
        if m.group("fn"):
            cur = {}
            out.append((m.group("fn"), cur))
        elif cur is not None:
            cur[m.group("key").split(" ")[0]] = int(m.group("val"))
    return out


def check_resources(src: str, remarks: str):
    """Scratch in a ring kernel is an error (see the module docstring).  Returns (table lines, error lines)."""
    lines, errors = [], []
    for fn, r in kernel_resources(remarks):
        sc = r.get("ScratchSize", 0)
        lines.append(f"{src:16s} vgpr {r.get('VGPRs', 0):3d} agpr {r.get('AGPRs', 0):3d} scratch {sc:4d} occ {r.get('Occupancy', 0)} lds {r.get('LDS', 0):6d}  {fn}")
        if sc and any(k in fn for k in RING_KERNELS):
            errors.append(f"{src}: {fn} needs {sc} bytes/lane of scratch: an LDS-DMA ring kernel must not spill (its counted vmcnt waits do not know of scratch traffic)")
    return lines, errors


def build(force=False, verbose=True, diag=False):
    lib = lib_path(diag)
    if not force and not needs_build(diag):
        return lib
    hipcc = _hipcc()
    od = _objdir(diag)
    os.makedirs(od, exist_ok=True)
    flags = FLAGS + (["-DMOY_DIAG=1"] if diag else [])
    objs, procs = [], []
    hdr_t = max(os.path.getmtime(d) for d in _deps() if not d.endswith(".hip"))
    for src in SOURCES:
        path = os.path.join(CSRC, src)
        obj = os.path.join(od, src.replace(".hip", ".o"))
        rem = obj + ".remarks.txt"
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.exists(rem) and os.path.getmtime(obj) > max(os.path.getmtime(path), hdr_t):
            continue
        cmd = [hipcc, *flags, "-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        # the translation units are independent: compile them side by side; the remarks (stderr) go to a file beside the object
        procs.append((cmd, src, rem, subprocess.Popen(cmd, stderr=open(rem + ".tmp", "w"))))
    for cmd, src, rem, p in procs:
        rc = p.wait()
        text = open(rem + ".tmp").read()
        os.replace(rem + ".tmp", rem)
        if rc != 0:
            sys.stderr.write(text[-8000:])
            os.remove(rem)
            raise subprocess.CalledProcessError(rc, cmd)
        diags = [ln for ln in text.splitlines() if ("warning:" in ln or "error:" in ln)]
        if diags and verbose:
            print("\n".join(diags[:40]), flush=True)
    table, errors = [], []
    for src, obj in zip(SOURCES, objs):
        ls, es = check_resources(src, open(obj + ".remarks.txt").read())
        table += ls
        errors += es
    with open(os.path.join(od, "resources.txt"), "w") as f:
        f.write("\n".join(table) + "\n")
    if errors:
        raise RuntimeError("kernel resource check failed:\n  " + "\n  ".join(errors))
    spilled = [ln for ln in table if " scratch    0 " not in ln]
    if verbose and spilled:
        print(f"[build] {len(spilled)} kernel(s) with scratch (none of them an LDS-DMA ring kernel): see {os.path.join(od, 'resources.txt')}", flush=True)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    force = "--force" in sys.argv
    if "--all" in sys.argv:
        build(force=force)
        build(force=force, diag=True)
    else:
        build(force=force, diag="--diag" in sys.argv)

#!/usr/bin/env python3
"""CU-partitioned concurrency with CU-MASKED streams ("the query-sized chain of one engine on a CU subset beside
the other engine's full-chip convs, instead of two streams of full-chip persistent kernels that serialise").

hipExtStreamCreateWithCUMask gives a stream whose dispatches only land on the compute units of its mask; moy_set_cu_limit sizes the
persistent kernels' grids to that many units.  Two engines (C2, bf16, `--frames` each, frames resident), eager launches, steady
state over `--batches` batches:

  free      today's mechanism: engine A on one plain stream, engine B on another, both free running on the whole chip
  halves    engine A on units [0, 128), engine B on [128, 256): two independent half chips
  pipe K    front of the plan (backbone, neck, value planes, score pass: launches [0, split)) of batch n on the 256 - K units of
            stream X, while the query-sized chain (top-k, decoder layers, assignment: [split, end)) of batch n - 1 runs on the K
            units of stream Y; batches alternate between the two engines' buffers

    python tools/probes/cu_mask_pipeline.py --out gpurun_out/cu_mask.json
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mo_yolo_amd.engine import TrackEngine  # noqa: E402
from mo_yolo_amd.fixtures import fixture  # noqa: E402
from mo_yolo_amd.synth import SyntheticSequence  # noqa: E402

hip = C.CDLL("libamdhip64.so")


def masked_stream(lo, hi, total):
    """Stream on the units whose mask bit index lies in [lo, hi).  (KFD deals consecutive mask bits round-robin over the XCDs and,
    inside one, over its shader engines, so a contiguous bit range is an equal share of every XCD.)"""
    words = (total + 31) // 32
    arr = (C.c_uint32 * words)()
    for i in range(lo, hi):
        arr[i // 32] |= 1 << (i % 32)
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), C.c_uint32(words), arr)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask rc {rc}")
    return torch.cuda.ExternalStream(st.value)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=288)
    ap.add_argument("--batches", type=int, default=12)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    cfg, arch, sd = fixture("c2")
    H, W, B = cfg["H"], cfg["W"], a.frames
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    engs = [TrackEngine(arch, sd, H, W, batch=B, dtype=torch.bfloat16) for _ in range(2)]
    seq = SyntheticSequence(0, H, W, cfg["style"])
    fr = torch.from_numpy(np.concatenate([seq.frames(t, 1) for t in range(8)])).cuda()
    for e in engs:
        e.inputs[0].copy_(fr[torch.arange(B, device="cuda") % 8])
        e.forward()
    torch.cuda.synchronize()
    lib = engs[0].lib
    split = engs[0]._split
    doc = {"frames_per_batch": B, "batches": a.batches, "compute_units": ncu, "split_launch": split, "launches": engs[0].num_launches,
           "runs": {}}

    def report(name, ms_total):
        per = ms_total / a.batches
        doc["runs"][name] = {"ms_per_batch": round(per, 3), "frames_per_s": round(B / per * 1e3, 1)}
        print(name, doc["runs"][name], flush=True)

    def wall(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3

    # ---- free: two plain streams
    plain = [torch.cuda.Stream(), torch.cuda.Stream()]

    def free(streams, limit):
        def go():
            lib.moy_set_cu_limit(limit)
            for n in range(a.batches // 2):
                for e, s in zip(engs, streams):
                    with torch.cuda.stream(s):
                        e.run_steps()
            lib.moy_set_cu_limit(0)
        return go

    wall(free(plain, 0))
    report("free: two plain streams, whole chip each", wall(free(plain, 0)))
    report("free (repeat)", wall(free(plain, 0)))

    def one(stream, limit):
        def go():
            lib.moy_set_cu_limit(limit)
            with torch.cuda.stream(stream):
                for n in range(a.batches):
                    engs[n % 2].run_steps()
            lib.moy_set_cu_limit(0)
        return go

    wall(one(plain[0], 0))
    report("one plain stream, whole chip", wall(one(plain[0], 0)))

    half = ncu // 2
    hs = [masked_stream(0, half, ncu), masked_stream(half, ncu, ncu)]
    wall(free(hs, half))
    report(f"halves: two masked streams of {half} units, grids sized {half}", wall(free(hs, half)))
    wall(one(hs[0], half))
    report(f"one masked stream of {half} units alone (the other half idle)", wall(one(hs[0], half)))

    # ---- pipe K
    for K in (32, 48, 64, 96):
        X, Y = masked_stream(0, ncu - K, ncu), masked_stream(ncu - K, ncu, ncu)

        def pipe():
            ev_front = [None, None]
            ev_back = [None, None]
            for n in range(a.batches):
                e = engs[n % 2]
                with torch.cuda.stream(X):
                    if ev_back[n % 2] is not None:
                        X.wait_event(ev_back[n % 2])           # the engine's buffers are free again
                    lib.moy_set_cu_limit(ncu - K)
                    e.run_steps(0, split)
                    ev_front[n % 2] = X.record_event()
                with torch.cuda.stream(Y):
                    Y.wait_event(ev_front[n % 2])
                    lib.moy_set_cu_limit(K)
                    e.run_steps(split)
                    ev_back[n % 2] = Y.record_event()
            lib.moy_set_cu_limit(0)

        wall(pipe)
        report(f"pipe: front on {ncu - K} units | query chain on {K} units", wall(pipe))
        # the two parts alone on their share (what bounds the pipe)
        def front_only():
            lib.moy_set_cu_limit(ncu - K)
            with torch.cuda.stream(X):
                for n in range(a.batches):
                    engs[n % 2].run_steps(0, split)
            lib.moy_set_cu_limit(0)

        def back_only():
            lib.moy_set_cu_limit(K)
            with torch.cuda.stream(Y):
                for n in range(a.batches):
                    engs[n % 2].run_steps(split)
            lib.moy_set_cu_limit(0)

        wall(front_only)
        report(f"  front alone on {ncu - K} units", wall(front_only))
        wall(back_only)
        report(f"  query chain alone on {K} units", wall(back_only))

    def front_full():
        with torch.cuda.stream(plain[0]):
            for n in range(a.batches):
                engs[n % 2].run_steps(0, split)

    def back_full():
        with torch.cuda.stream(plain[0]):
            for n in range(a.batches):
                engs[n % 2].run_steps(split)

    wall(front_full)
    report("front alone, whole chip", wall(front_full))
    wall(back_full)
    report("query chain alone, whole chip", wall(back_full))
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(doc, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()

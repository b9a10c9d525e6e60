#!/usr/bin/env python3
"""CU-partition measurement (first bullet): does a bandwidth-bound launch of the plan keep its rate on a PART of the
chip, and does a matrix-rate-bound launch of the other stream then use the rest?

  1. each launch ALONE at MOY_CU_LIMIT = 256, 192, 160, 128, 96, 64 compute units (the persistent kernels size their grids by it):
       value    value projection of the P3 level, M = 288 x 10336, K = 128 -> 1536 columns in head planes (write-bound, 9.9 GB)
       gemm128  1x1 conv 128 -> 128 at the P3 level (read + write, 1.5 GB)
       conv128  3x3 conv 128 -> 128 at the P4 level, M = 744192 (matrix-rate bound)
  2. PAIRS on two streams: `value` on L units beside a chain of `conv128` launches on 256 - L units, against the same work
     back to back on the whole chip -- and against both on 256 units on two streams (what the two engines of bench.py do today).

    python tools/probes/cu_share.py [--out gpurun_out/cu_share.json]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from mo_yolo_amd import ops  # noqa: E402

DEV, DT = "cuda", torch.bfloat16


def limit(n):
    if n is None or n >= 256:
        os.environ.pop("MOY_CU_LIMIT", None)
    else:
        os.environ["MOY_CU_LIMIT"] = str(n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--frames", type=int, default=288)
    ap.add_argument("--value-only", action="store_true", help="time the value launch alone on the whole chip and exit (A/B of MOY_WREG_V128 forms)")
    a = ap.parse_args()
    B = a.frames
    g = torch.Generator(device="cpu").manual_seed(1)

    def rnd(*s, scale=1.0):
        return (torch.rand(*s, generator=g) - 0.5) * 2 * scale

    # value form (P3 level of the C2 plan)
    hw3, S = 76 * 136, 13566
    x3 = rnd(B * hw3, 128).to(DEV, DT)
    wv = ops.pad_weight(rnd(1536, 128, scale=0.1).to(DEV), DT)
    bv = rnd(1536, scale=0.1).to(DEV)
    planes = torch.zeros(48, B * S, 32, device=DEV, dtype=DT)

    def value():
        ops.gemm(x3, wv, 1536, 128, out=planes[0, :B * hw3], shift=bv, planes=(32, B * S * 32), c_rpb=hw3, c_bstride=S)

    if a.value_only:
        value()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ts = []
        for _ in range(5):
            e0.record()
            for _ in range(10):
                value()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10)
        print("value_ms", os.environ.get("MOY_WREG_V128", "0"), [round(t, 4) for t in ts], flush=True)
        return
    w128 = ops.pad_weight(rnd(128, 128, scale=0.1).to(DEV), DT)
    sc, sh = (rnd(128) * 0.2 + 1).to(DEV), rnd(128, scale=0.1).to(DEV)
    y3 = torch.empty(B * hw3, 128, device=DEV, dtype=DT)

    def gemm128():
        ops.gemm(x3, w128, 128, 128, out=y3, scale=sc, shift=sh, act=1)

    h4, w4 = 38, 68
    x4 = rnd(B * h4 * w4, 128).to(DEV, DT)
    wc = ops.pad_weight(rnd(128, 1152, scale=0.03).to(DEV), DT)
    y4 = torch.empty(B * h4 * w4, 128, device=DEV, dtype=DT)

    def conv128():
        ops.gemm(x4, wc, 128, 1152, out=y4, ksize=3, stride=1, geom=(B, h4, w4, h4, w4, 128), scale=sc, shift=sh, act=1)

    def timed(fn, reps=10):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    doc = {"frames": B, "alone_ms": {}, "pairs": []}
    for name, fn in (("value", value), ("gemm128", gemm128), ("conv128", conv128)):
        row = {}
        for n in (256, 224, 192, 160, 128, 96, 64):
            limit(n)
            row[str(n)] = round(timed(fn), 4)
        doc["alone_ms"][name] = row
        print(name, row, flush=True)
    limit(None)

    # pairs: value on L units || k x conv128 on 256 - L units
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()

    def pair(L, k, reps=6):
        def once():
            limit(L)
            with torch.cuda.stream(sA):
                value()
            limit(256 - L if L < 256 else None)
            with torch.cuda.stream(sB):
                for _ in range(k):
                    conv128()
        for _ in range(2):
            once()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        cur = torch.cuda.current_stream()
        e0.record(cur)
        sA.wait_stream(cur)
        sB.wait_stream(cur)
        for _ in range(reps):
            once()
        cur.wait_stream(sA)
        cur.wait_stream(sB)
        e1.record(cur)
        torch.cuda.synchronize()
        limit(None)
        return e0.elapsed_time(e1) / reps

    tv, tc = doc["alone_ms"]["value"]["256"], doc["alone_ms"]["conv128"]["256"]
    for k in (4, 6, 8):
        serial = tv + k * tc
        for L in (256, 192, 160, 128, 96, 64):
            t = pair(L, k)
            doc["pairs"].append({"value_units": L, "conv_units": 256 - L if L < 256 else 256, "conv_launches": k, "ms": round(t, 4),
                                 "back_to_back_whole_chip_ms": round(serial, 4), "ratio": round(t / serial, 4)})
            print(doc["pairs"][-1], flush=True)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(doc, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()

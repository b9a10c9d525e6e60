#!/usr/bin/env python3
"""Probe (round 6): four live sequences per step -- ONE engine of 4 frames on one stream against StreamedEngines of 2 x 2 and 4 x 1 frames on 2 / 4
streams (hipGraph replay each), device synchronised after every step.  C5 fixture (fp16)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from mo_yolo_amd.engine import StreamedEngines, TrackEngine
from mo_yolo_amd.fixtures import fixture
from mo_yolo_amd.synth import SyntheticSequence

cfg, arch, sd = fixture("c2")
H, W = cfg["H"], cfg["W"]
dt = torch.float16 if os.environ.get("SB_DT", "f16") == "f16" else torch.bfloat16
B = int(os.environ.get("SB_B", 4))
seqs = [SyntheticSequence(i, H, W, cfg["style"]) for i in range(B)]
frames = torch.cat([s.frames_torch(8, 1, device="cuda") for s in seqs])


def timed(step, n=300, warm=30):
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    ls = []
    for _ in range(n):
        t = time.perf_counter()
        step()
        torch.cuda.synchronize()
        ls.append(time.perf_counter() - t)
    ls.sort()
    return sum(ls) / len(ls) * 1e3, ls[len(ls) // 2] * 1e3, ls[int(len(ls) * 0.99)] * 1e3


e1 = TrackEngine(arch, sd, H, W, batch=B, dtype=dt)
e1.inputs[0].copy_(frames)
e1.forward(slot=0); torch.cuda.synchronize(); e1.capture()
ref = {k: v.clone() for k, v in e1.outputs().items() if hasattr(v, "shape")}
print(f"1 engine x {B} frames, 1 stream : mean %.4f p50 %.4f p99 %.4f ms" % timed(lambda: e1.forward(slot=0)))
for S in (2, 4):
    if B % S:
        continue
    se = StreamedEngines(arch, sd, H, W, batch=B, streams=S, dtype=dt)
    se.load(frames, 0)
    se.forward(None, slot=0); torch.cuda.synchronize()
    m = timed(lambda: se.forward(None, slot=0))
    same = all(torch.equal(torch.cat([e.outputs()["obj_idxes"] for e in se.engines]), ref["obj_idxes"]) for _ in (0,))
    print(f"{S} engines x {B // S} frames, {S} streams: mean %.4f p50 %.4f p99 %.4f ms   (ids equal to the one-engine run: {same})" % m)

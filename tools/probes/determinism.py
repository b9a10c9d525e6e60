#!/usr/bin/env python3
"""Run-to-run determinism of an engine: the same frames through the same plan N times; every layer output, the value planes,
the score logits and the results must be bit-identical from pass to pass (no atomics anywhere in the forward path).  Prints the
first buffer that differs, how many elements and where -- a kernel with a race shows up as the first differing layer.

    python tools/probes/determinism.py [--dtype f16] [--batch 104] [--passes 20] [--config c2]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mo_yolo_amd.engine import TrackEngine  # noqa: E402
from mo_yolo_amd.fixtures import fixture  # noqa: E402
from mo_yolo_amd.synth import SyntheticSequence  # noqa: E402


def snapshot(e):
    d = {}
    for i, v in sorted(e.layer_views.items()):
        if v is not None and i not in e.virtual_layers:
            d[f"layer{i:02d}"] = v.tensor().clone()
    d["value_planes"] = e.value_planes.clone()
    d["scores_all"] = e.scores_all.clone()
    for k, v in e.outputs().items():
        if hasattr(v, "clone"):
            d["out." + k] = v.clone()
    return d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--batch", type=int, default=104)
    ap.add_argument("--passes", type=int, default=20)
    ap.add_argument("--config", default="c2")
    ap.add_argument("--graph", action="store_true")
    a = ap.parse_args()
    dt = {"f16": torch.float16, "bf16": torch.bfloat16}[a.dtype]
    cfg, arch, sd = fixture(a.config)
    H, W, B = cfg["H"], cfg["W"], a.batch
    seq = SyntheticSequence(0, H, W, cfg["style"])
    fr = torch.from_numpy(np.concatenate([seq.frames(t, 1) for t in range(8)])).cuda()
    fr = fr[torch.arange(B, device="cuda") % 8].contiguous()
    e = TrackEngine(arch, sd, H, W, batch=B, dtype=dt)
    print("fold", getattr(e, "fold_proj", None), "launches", e.num_launches, "virtual layers", sorted(e.virtual_layers), flush=True)
    e.forward(fr)
    if a.graph:
        e.capture()
        e.forward(fr)
    torch.cuda.synchronize()
    ref = snapshot(e)
    bad = 0
    for n in range(a.passes):
        # disturb the caches / the timing between passes a little: another engine-sized allocation being filled
        junk = torch.empty(64 << 20, device="cuda", dtype=torch.uint8).random_()
        e.forward(fr)
        torch.cuda.synchronize()
        cur = snapshot(e)
        diffs = [(k, int((cur[k] != ref[k]).sum())) for k in ref if not torch.equal(cur[k], ref[k])]
        del junk
        if diffs:
            bad += 1
            k0, n0 = diffs[0]
            idx = (cur[k0] != ref[k0]).nonzero()
            print(f"pass {n}: {len(diffs)} buffers differ; first {k0}: {n0} elements, first at {idx[0].tolist()} last at {idx[-1].tolist()} "
                  f"shape {list(ref[k0].shape)}; all: {diffs[:8]}", flush=True)
    print(f"{a.dtype} B={B}: {bad} of {a.passes} passes differ from the first")


if __name__ == "__main__":
    main()

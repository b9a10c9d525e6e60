#!/usr/bin/env python3
"""Probe: the split-fp16 stem (moy_stem_conv_x3) and the scalar fp32 stem on B frames of 608 x 1088 -- time per launch; run it under
rocprofv3 --pmc for the counters (tools/gpu/r06s.sh)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from mo_yolo_amd import ops

B = int(os.environ.get("SX_B", 96))
g = torch.Generator().manual_seed(0)
u8 = torch.randint(0, 256, (B, 608, 1088, 3), generator=g, dtype=torch.uint8).cuda()
w = (torch.rand(32, 3, 3, 3, generator=g) - 0.5) * 0.6
sc, sh = (torch.rand(32, generator=g) * 0.2 + 0.9).cuda(), ((torch.rand(32, generator=g) - 0.5) * 0.2).cuda()
w3 = ops.stem_weights_x3(w.cuda())
w27 = w.permute(2, 3, 1, 0).reshape(27, 32).contiguous().cuda()
for name, f in (("stem_x3", lambda: ops.stem_conv_x3(u8, w3, sc, sh)), ("stem fp32", lambda: ops.stem_conv(u8, w27, sc, sh, torch.float32))):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"{name:10s} {ms * 1e3:8.1f} us   {(u8.numel() + B * 304 * 544 * 32 * 4) / ms / 1e6:7.0f} GB/s")

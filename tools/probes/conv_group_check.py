#!/usr/bin/env python3
"""3x3 stride-1 convs at C = 128 on image sizes whose 16-column tiling wastes columns (68 -> 5 tiles, 85 %): outputs of the
weight-stationary kernel to a file, so that the grouped tiling (G images on one virtual row, csrc/conv_ws.hip GRP forms) can be
compared bit for bit with the plain tiling of another process (MOY_CWS_GROUP=0).   usage: conv_group_check.py <out.pt> [dtype]"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from mo_yolo_amd import _lib as L  # noqa: E402
from mo_yolo_amd import ops  # noqa: E402

SHAPES = [(40, 38, 68, False), (41, 38, 68, True), (44, 37, 67, True), (50, 40, 70, False)]


def run(dt):
    outs = []
    for i, (B, H, W, res) in enumerate(SHAPES):
        g = torch.Generator().manual_seed(100 + i)
        C = 128
        x = ((torch.rand(B * H * W, C, generator=g) - 0.5) * 2).to(dt)
        w = ((torch.rand(C, 9 * C, generator=g) - 0.5) * 2 / math.sqrt(9 * C)).to(dt)
        sc, sh = torch.rand(C, generator=g) * 0.2 + 0.9, (torch.rand(C, generator=g) - 0.5) * 0.2
        r = ((torch.rand(B * H * W, C, generator=g) - 0.5) * 2).to(dt)
        buf = torch.zeros(B * H * W, 3 * C + 8, device="cuda", dtype=dt)           # input | residual | output as channel slices
        buf[:, :C] = x.cuda()
        buf[:, C:2 * C] = r.cuda()
        ops.gemm(buf[:, :C], ops.pad_weight(w.cuda(), dt), C, 9 * C, ksize=3, stride=1, geom=(B, H, W, H, W, C), scale=sc.cuda(), shift=sh.cuda(),
                 act=L.ACT_SILU, R=buf[:, C:2 * C] if res else None, out=buf[:, 2 * C:3 * C])
        torch.cuda.synchronize()
        assert float(buf[:, 3 * C:].abs().max()) == 0 and torch.equal(buf[:, :C].cpu(), x) and torch.equal(buf[:, C:2 * C].cpu(), r)
        outs.append(dict(shape=(B, H, W, res), x=x, w=w, sc=sc, sh=sh, r=r, y=buf[:, 2 * C:3 * C].cpu().clone()))
    return outs


if __name__ == "__main__":
    dt = torch.float16 if (len(sys.argv) > 2 and sys.argv[2] == "f16") else torch.bfloat16
    torch.save(run(dt), sys.argv[1])
    print("saved", sys.argv[1], "MOY_CWS_GROUP =", os.environ.get("MOY_CWS_GROUP", "1"))

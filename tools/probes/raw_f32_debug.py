"""Debug probe (round 6): where the fp32 raw gather differs from the oracle -- by head, by channel group, by query position in its block."""
import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mo_yolo_amd import ops
from oracle import track_oracle as O
DEV = "cuda:0"
def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale
for (B, Lq, shapes, sel) in ((1, 24, [(9, 7)], None), (1, 24, [(9, 7)], 0), (1, 24, [(9, 7)], 64), (2, 37, [(12, 20), (6, 10), (3, 5)], None)):
    H0, W0 = shapes[0]; nl = len(shapes); S1 = sum(h * w for h, w in shapes[1:])
    x = rnd(B * H0 * W0, 128, seed=1)
    wc = rnd(256, 128, seed=2, scale=1 / math.sqrt(128)); bc = rnd(256, seed=3, scale=0.5)
    if sel is not None:           # W_h = selector of channels sel..sel+31, no bias: the output IS the gathered vector
        wc = torch.zeros(256, 128); bc = torch.zeros(256)
        for h in range(8):
            for c in range(32):
                wc[h * 32 + c, sel + c] = 1.0
    v1 = rnd(B, max(S1, 1), 256, seed=4)
    offaw = torch.cat([rnd(B * Lq, 8 * nl * 8, seed=5, scale=2.0), rnd(B * Lq, 8 * nl * 4, seed=6, scale=2.0)], 1)
    ref_box = torch.rand(B * Lq, 4, generator=torch.Generator().manual_seed(7)); ref_box[:, 2:] = ref_box[:, 2:] * 0.5 + 0.05
    planes = v1.view(B * S1, 8, 32).permute(1, 0, 2).contiguous().to(DEV) if S1 else None
    y = ops.msda_raw0(x.to(DEV), wc.to(DEV), bc.to(DEV), planes, B, shapes, offaw.to(DEV), ref_box.to(DEV), Lq).cpu()
    v0 = (x @ wc.T + bc).view(B, H0 * W0, 256)
    value = torch.cat([v0, v1[:, :S1]], 1) if S1 else v0
    off = offaw[:, :8 * nl * 8].view(B, Lq, 8, nl, 4, 2)
    aw = torch.softmax(offaw[:, 8 * nl * 8:].view(B, Lq, 8, nl * 4), -1).view(B, Lq, 8, nl, 4)
    rb = ref_box.view(B, Lq, 1, 1, 1, 4)
    loc = rb[..., :2] + off / 4 * rb[..., 2:] * 0.5
    want = O.msda_core(value.view(B, -1, 8, 32), shapes, loc, aw).view(B * Lq, 256)
    e = (y - want).abs()
    print(f"B{B} Lq{Lq} L{nl} sel{sel}: max err {float(e.max()):.4g}")
    print("  by head      ", [round(float(e[:, h * 32:(h + 1) * 32].max()), 4) for h in range(8)])
    print("  by channel%32", [round(float(e.view(-1, 8, 32)[:, :, c].max()), 3) for c in range(0, 32, 2)])
    print("  by query%8   ", [round(float(e[qq::8].max()), 4) for qq in range(8)])
    if sel is not None:
        print("  row 3 head 0 got ", [round(float(v), 4) for v in y[3, :8]]); print("  row 3 head 0 want", [round(float(v), 4) for v in want[3, :8]])

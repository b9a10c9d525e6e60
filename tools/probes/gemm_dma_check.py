#!/usr/bin/env python3
"""Where does the large-tile DMA kernel (csrc/gemm_dma.hip) differ from the tiled kernel?  Mismatch map by row tile / column / image."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mo_yolo_amd import _lib as L, ops
DEV = "cuda"
def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale
def run(Cin, Cout, s, B, H, W, res, dt=torch.bfloat16, per=None):
    x = rnd(B, Cin, H, W, seed=1).to(dt)
    w = rnd(Cout, Cin, 3, 3, seed=2, scale=1 / math.sqrt(9 * Cin)).to(dt)
    sc, sh = rnd(Cout, seed=3) * 0.2 + 1, rnd(Cout, seed=4, scale=0.1)
    Ho, Wo = (H + 2 - 3) // s + 1, (W + 2 - 3) // s + 1
    M = B * Ho * Wo
    xin = x.permute(0, 2, 3, 1).reshape(B * H * W, Cin).contiguous().to(DEV)
    rs = rnd(M, Cout, seed=5).to(dt).to(DEV) if res else None
    wp = ops.pad_weight(w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).float().to(DEV), dt)
    kw = dict(ksize=3, stride=s, scale=sc.to(DEV), shift=sh.to(DEV), act=L.ACT_SILU)
    out = torch.empty(M, Cout, device=DEV, dtype=dt)
    ops.gemm(xin, wp, Cout, 9 * Cin, geom=(B, H, W, Ho, Wo, Cin), R=rs, out=out, **kw)
    two = torch.empty(M, Cout, device=DEV, dtype=dt)
    tn = Cout // (256 if Cout % 256 == 0 else 128)
    per = per or max(1, 300 * 256 // tn // (Ho * Wo))
    for b0 in range(0, B, per):
        b1 = min(B, b0 + per)
        ops.gemm(xin[b0 * H * W:b1 * H * W], wp, Cout, 9 * Cin, geom=(b1 - b0, H, W, Ho, Wo, Cin),
                 R=rs[b0 * Ho * Wo:b1 * Ho * Wo] if res else None, out=two[b0 * Ho * Wo:b1 * Ho * Wo], **kw)
    torch.cuda.synchronize()
    bad = (out != two)
    n = int(bad.sum())
    print(f"Cin {Cin} Cout {Cout} s{s} B{B} {H}x{W} res {res}: M {M}, mismatches {n} of {out.numel()}")
    if n:
        rows = bad.any(1).nonzero().flatten()
        cols = bad.any(0).nonzero().flatten()
        print("  rows with mismatches:", len(rows), "first", rows[:12].tolist(), "last", rows[-5:].tolist())
        print("  row % 256 histogram (16 bins):", torch.bincount((rows % 256) // 16, minlength=16).tolist())
        print("  tiles affected:", len(torch.unique(rows // 256)), "of", (M + 255) // 256, " images:", len(torch.unique(rows // (Ho * Wo))))
        print("  pixel-in-image of first rows:", [(int(r) % (Ho * Wo)) for r in rows[:12]])
        print("  cols with mismatches:", len(cols), cols[:16].tolist())
        d = (out.float() - two.float()).abs()
        print("  max abs diff", float(d.max()), " mean |two|", float(two.float().abs().mean()))
    return n
if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "n128":       # the N = 128 shapes (with MOY_GEMM_DMA_FORM=1|2: the 512 x 128 / 256 x 128 forms)
        # big launch: >= 384 tiles in either form; reference launches: < 384 tiles of 256 rows -> the tiled kernel
        run(128, 128, 2, 80, 75, 135, False, per=300 * 256 // (38 * 68))
        run(192, 128, 1, 20, 76, 136, True, per=300 * 256 // (76 * 136))
        run(128, 128, 1, 310, 19, 34, True, per=300 * 256 // (19 * 34), dt=torch.float16)
        sys.exit(0)
    run(256, 256, 1, 156, 19, 34, True)
    run(256, 256, 1, 156, 19, 34, False)
    run(256, 256, 1, 156, 19, 34, False, dt=torch.float16)
    run(128, 256, 1, 40, 38, 68, False)
    run(256, 256, 2, 156, 38, 68, False)

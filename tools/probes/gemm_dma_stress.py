#!/usr/bin/env python3
"""Randomised equality sweep of the large-tile LDS-DMA GEMM (csrc/gemm_dma.hip) against the tiled kernel: random channel counts, image
sizes, strides, residual on / off, 3x3 and 1x1, every case launched several times (a race in the hand-counted DMA pipeline would show
as a tile that differs on SOME launch).  The reference = the same rows submitted as launches below the 384-tile threshold."""
import math, os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mo_yolo_amd import _lib as L, ops
DEV = "cuda"
rng = random.Random(int(os.environ.get("GD_SEED", 0)))
n_cases, reps = int(os.environ.get("GD_CASES", 24)), int(os.environ.get("GD_REPS", 4))
bad_total = 0
for case in range(n_cases):
    dt = rng.choice([torch.bfloat16, torch.float16])
    N = rng.choice([256, 256, 512])
    res = rng.random() < 0.5
    act = rng.choice([L.ACT_SILU, L.ACT_SILU, L.ACT_NONE, L.ACT_RELU])
    g = torch.Generator().manual_seed(case)
    if rng.random() < 0.7:
        Cin = rng.choice([64, 128, 192, 256, 320])
        s = rng.choice([1, 2])
        H, W = rng.randint(9, 60), rng.randint(9, 90)
        Ho, Wo = (H + 2 - 3) // s + 1, (W + 2 - 3) // s + 1
        need = 384 * 256 // (N // 256)
        B = need // (Ho * Wo) + rng.randint(1, 3)
        M, K = B * Ho * Wo, 9 * Cin
        x = ((torch.rand(B * H * W, Cin, generator=g) - 0.5)).to(dt).to(DEV)
        geom = dict(ksize=3, stride=s, geom=(B, H, W, Ho, Wo, Cin))
        per = max(1, (300 * 256 // (N // 256)) // (Ho * Wo))
        chunks = [(b0, min(B, b0 + per)) for b0 in range(0, B, per)]
        sub = lambda b0, b1: (x[b0 * H * W:b1 * H * W], dict(ksize=3, stride=s, geom=(b1 - b0, H, W, Ho, Wo, Cin)), slice(b0 * Ho * Wo, b1 * Ho * Wo))
        desc = f"3x3 s{s} {Cin}->{N} {B}x{H}x{W}"
    else:
        K = rng.choice([512, 640, 1024, 1280])
        M = 384 * 256 // (N // 256) + rng.randint(1, 5000)
        x = ((torch.rand(M, K, generator=g) - 0.5)).to(dt).to(DEV)
        geom = {}
        step = 300 * 256 // (N // 256)
        chunks = [(m0, min(M, m0 + step)) for m0 in range(0, M, step)]
        sub = lambda m0, m1: (x[m0:m1], {}, slice(m0, m1))
        desc = f"1x1 {K}->{N} M{M}"
    w = ops.pad_weight(((torch.rand(N, K, generator=g) - 0.5) / math.sqrt(K)).to(DEV), dt)
    sc, sh = (torch.rand(N, generator=g) * 0.4 + 0.8).to(DEV), ((torch.rand(N, generator=g) - 0.5) * 0.2).to(DEV)
    r = ((torch.rand(M, N, generator=g) - 0.5)).to(dt).to(DEV) if res else None
    kw = dict(scale=sc, shift=sh, act=act)
    ref = torch.empty(M, N, device=DEV, dtype=dt)
    for c0, c1 in chunks:
        xs, gk, rows = sub(c0, c1)
        ops.gemm(xs, w, N, K, out=ref[rows], R=r[rows] if res else None, **gk, **kw)
    nbad = 0
    for _ in range(reps):
        out = torch.empty(M, N, device=DEV, dtype=dt)
        ops.gemm(x, w, N, K, out=out, R=r, **geom, **kw)
        torch.cuda.synchronize()
        nbad += int((out != ref).sum())
    bad_total += nbad
    print(f"case {case:3d} {str(dt)[6:]:9s} {desc:34s} res {int(res)} act {act}: {'equal' if nbad == 0 else f'{nbad} VALUES DIFFER'} ({reps} launches)", flush=True)
print("ALL EQUAL" if bad_total == 0 else f"FAILED: {bad_total} values differ")
sys.exit(0 if bad_total == 0 else 1)

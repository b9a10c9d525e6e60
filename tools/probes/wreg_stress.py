#!/usr/bin/env python3
"""Randomised equality sweep of the weight-stationary kernel's forms against the tiled kernel (the same rows submitted as launches
below 65536 rows): seeded (accumulator seed fetched by hand-counted asynchronous loads), K = 192, head planes, row remap.  Every
case is launched several times: a counted-wait race would show as a run-to-run difference.  usage: wreg_stress.py [cases=24] [seed=0]"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mo_yolo_amd import ops, _lib as L
dev = "cuda"
ncases, seed = (int(sys.argv[1]) if len(sys.argv) > 1 else 24), (int(sys.argv[2]) if len(sys.argv) > 2 else 0)
rng = random.Random(seed)
torch.manual_seed(seed)
bad = 0
for case in range(ncases):
    dt = rng.choice([torch.bfloat16, torch.float16])
    kind = rng.choice(["seed128", "seed256", "k192", "planes", "plain256", "n128k256"])
    H, W = 2 * rng.randint(8, 40), 2 * rng.randint(8, 70)
    B = (65536 // (H * W) + rng.randint(1, 3)) * int(os.environ.get("WS_SCALE", 1))   # WS_SCALE: long row-tile sequences per block
    M = B * H * W
    if kind == "seed128": N, K = 128, 128
    elif kind == "seed256": N, K = 256, 256
    elif kind == "k192": N, K = 128, 192
    elif kind == "planes": N, K = 512, 256
    elif kind == "plain256": N, K = 256, rng.choice([128, 256, 384, 512])
    else: N, K = 128, 256
    x = ((torch.rand(M, K, device=dev) - 0.5)).to(dt)
    w = ops.pad_weight((torch.rand(N, K, device=dev) - 0.5) / K ** 0.5, dt)
    sc, sh = torch.rand(N, device=dev) + 0.5, torch.rand(N, device=dev) - 0.5
    kw = dict(scale=sc, shift=sh, act=L.ACT_SILU)
    hw = (H // 2) * (W // 2)
    seedt = (torch.rand(B * hw, N, device=dev) - 0.5) if kind.startswith("seed") else None
    per = max(1, 65535 // (H * W))
    def run(big):
        if kind == "planes":
            out = torch.zeros(N // 32, M, 32, device=dev, dtype=dt)
            if big:
                ops.gemm(x, w, N, K, out=out[0], planes=(32, M * 32), **kw)
            else:
                for b0 in range(0, B, per):
                    r0, r1 = b0 * H * W, min(B, b0 + per) * H * W
                    tmp = torch.zeros(N // 32, r1 - r0, 32, device=dev, dtype=dt)
                    ops.gemm(x[r0:r1], w, N, K, out=tmp[0], planes=(32, (r1 - r0) * 32), **kw)
                    out[:, r0:r1] = tmp
            return out
        out = torch.zeros(M, N, device=dev, dtype=dt)
        if big:
            ops.gemm(x, w, N, K, out=out, pre=(seedt, H, W) if seedt is not None else None, **kw)
        else:
            for b0 in range(0, B, per):
                b1 = min(B, b0 + per)
                ops.gemm(x[b0 * H * W:b1 * H * W], w, N, K, out=out[b0 * H * W:b1 * H * W],
                         pre=(seedt[b0 * hw:b1 * hw], H, W) if seedt is not None else None, **kw)
        return out
    ref = run(False)
    ok = True
    for rep in range(4):
        got = run(True)
        torch.cuda.synchronize()
        if not torch.equal(got, ref):
            ok = False
            nd = int((got != ref).sum())
            print(f"case {case} {kind} {dt} B{B} H{H} W{W} N{N} K{K} rep {rep}: {nd} elements differ")
    bad += not ok
    if ok: print(f"case {case:2d} {kind:9s} {str(dt)[6:]:9s} M={M:7d} ({B}x{H}x{W}) N={N} K={K}: equal x4")
print("FAILED" if bad else "all equal", bad)
sys.exit(1 if bad else 0)

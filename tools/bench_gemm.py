#!/usr/bin/env python3
"""Micro-benchmark of moy_gemm on the shapes of the C2 plan (B=32 frames): us, GB/s (algorithmic
bytes), TF/s per shape.  Interleaved rounds in one process (cdna guide rule 24)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mo_yolo_amd import _lib as L
from mo_yolo_amd import ops

B = int(os.environ.get("BG_B", 32))
# BG_DT: bf16 (default) | f16 | f32 | f32x3 (fp32 tensors, pre-split fp16 weights: the MOY_F32X3 form of the tiled kernel)
DT_NAME = os.environ.get("BG_DT", "bf16")
X3 = DT_NAME == "f32x3"
dt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32, "f32x3": torch.float32}[DT_NAME]
ESZ = 4 if dt == torch.float32 else 2
dev = "cuda"
# (name, ksize, stride, Hin, Win, Cin, Cout, ln)
SHAPES = [
    ("L1 3x3s2 32->64", 3, 2, 304, 544, 32, 64, False),
    ("L2m 3x3 32->32", 3, 1, 152, 272, 32, 32, False),
    ("L2cv2 1x1 96->64", 1, 1, 152, 272, 96, 64, False),
    ("L3 3x3s2 64->128", 3, 2, 152, 272, 64, 128, False),
    ("L4m 3x3 64->64", 3, 1, 76, 136, 64, 64, False),
    ("L4cv2 1x1 256->128", 1, 1, 76, 136, 256, 128, False),
    ("L4cv1 1x1 128->128", 1, 1, 76, 136, 128, 128, False),
    ("N18cv2 1x1 192->128", 1, 1, 76, 136, 192, 128, False),
    ("L6cv1 1x1 256->256", 1, 1, 38, 68, 256, 256, False),
    ("L6cv2 1x1 512->256", 1, 1, 38, 68, 512, 256, False),
    ("N15cv2 1x1 384->256", 1, 1, 38, 68, 384, 256, False),
    ("P3proj 1x1 128->256", 1, 1, 76, 136, 128, 256, False),
    ("L6m 3x3 128->128", 3, 1, 38, 68, 128, 128, False),
    ("L5 3x3s2 128->256", 3, 2, 76, 136, 128, 256, False),
    ("N19 3x3s2 128->128", 3, 2, 76, 136, 128, 128, False),
    ("L7 3x3s2 256->256", 3, 2, 38, 68, 256, 256, False),
    ("L8m 3x3 128->128 P5", 3, 1, 19, 34, 128, 128, False),
    ("value 1x1 256->1536", 1, 1, 1, 13566, 256, 1536, False),
    ("encout+ln 256->256", 1, 1, 1, 13566, 256, 256, True),
    ("dec ffn1 256->1024", 1, 1, 1, 300, 256, 1024, False),
    ("dec ffn2+ln 1024->256", 1, 1, 1, 300, 1024, 256, True),
    ("dec qk 256->512", 1, 1, 1, 300, 256, 512, False),
    ("dec lin 256->256", 1, 1, 1, 300, 256, 256, False),
    ("dec lin+ln 256->256", 1, 1, 1, 300, 256, 256, True),
    ("dec offaw 256->288", 1, 1, 1, 300, 256, 288, False),
]


def make(s):
    name, ks, st, H, W, Cin, Cout, ln = s
    x = (torch.rand(B * H * W, Cin, device=dev) - 0.5).to(dt)
    K = ks * ks * Cin
    w0 = (torch.rand(Cout, K, device=dev) - 0.5) / K ** 0.5
    w = ops.split_weight(w0) if X3 else ops.pad_weight(w0, dt)
    sc = torch.rand(Cout, device=dev) + 0.5
    sh = torch.rand(Cout, device=dev) - 0.5
    if ks == 3:
        Ho, Wo = (H + 2 - 3) // st + 1, (W + 2 - 3) // st + 1
        geom = (B, H, W, Ho, Wo, Cin)
        M = B * Ho * Wo
    else:
        geom, M = None, B * H * W
    out = torch.empty(M, Cout, device=dev, dtype=dt)
    conv = H > 1          # backbone convs carry BN + SiLU; the head / decoder linears only a bias
    kw = dict(out=out, ksize=ks, stride=st, geom=geom, scale=sc if conv else None, shift=sh,
              act=(L.ACT_SILU if conv else L.ACT_NONE))
    if ln:
        kw.update(ln=(sc, sh), act=L.ACT_NONE)
    alg = (x.numel() + w.numel() + out.numel()) * ESZ
    flops = 2 * M * Cout * K
    if X3:
        kw["split_f16"] = True
    f = lambda: ops.gemm(x, w, Cout, K, **kw)
    f.out = out
    return f, alg, flops, M


def main():
    only = os.environ.get("BG_ONLY")
    shapes = [s for s in SHAPES if not only or any(o in s[0] for o in only.split(","))]
    cases = [(s[0],) + make(s) for s in shapes]
    for _, f, *_ in cases:
        f()
    torch.cuda.synchronize()
    rounds, reps = int(os.environ.get('BG_ROUNDS', 5)), int(os.environ.get('BG_REPS', 10))
    best = {c[0]: 1e9 for c in cases}
    for _ in range(rounds):
        for name, f, alg, flops, M in cases:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                f()
            e1.record()
            torch.cuda.synchronize()
            best[name] = min(best[name], e0.elapsed_time(e1) / reps)
    tot = 0
    for name, f, alg, flops, M in cases:
        ms = best[name]
        tot += ms
        print(f"{name:26s} M={M:8d} {ms*1e3:8.1f} us {alg/ms/1e6:7.0f} GB/s {flops/ms/1e9:7.1f} TF/s")
        if os.environ.get("MOY_CWS_ABL") == "5" and int(os.environ.get("MOY_CWS_VARIANT", "0")) & 8:   # ping-pong form: work / barrier cycles per interval, group A (wave 0) and B (wave 4)
            d = f.out.view(torch.int64).flatten()[:32].cpu().tolist()
            for g, o in (("A", 0), ("B", 16)):
                n = max(d[o + 8], 1)
                print(f"    group {g}: work per interval " + " ".join(f"{v / n:.0f}" for v in d[o:o + 4]) + " | at the barrier " +
                      " ".join(f"{v / n:.0f}" for v in d[o + 4:o + 8]) + f"  (cycles per tile, {n} tiles)")
        elif os.environ.get("MOY_GD_DIAG") == "1" and "3x3" in name:      # diagnostic build of csrc/gemm_dma.hip: section stamps of block 8, waves 0 and 4
            d = f.out.view(torch.int64).flatten()[:16].cpu().tolist()
            for g, o in (("wave 0", 0), ("wave 4", 8)):
                nkk = max(d[o + 6], 1)
                names = ["reads+dma issue", "vmcnt wait", "barrier A", "mfma", "barrier B"]
                print(f"    {g}: per k-tile (4 phases): " + ", ".join(f"{nm} {v / nkk:.0f}" for nm, v in zip(names, d[o:o + 5])) +
                      f" | epilogue {d[o + 5]} | k-tiles {nkk}")
        elif os.environ.get("MOY_CWS_ABL") == "5":      # diagnostic build of csrc/conv_ws.hip: phase stamps of block 0, wave 0
            d = f.out.view(torch.int64).flatten()[:8].cpu().tolist()
            n = max(d[7], 1)
            names = ["setup+dma", "mfma", "epilogue", "barrier1", "stores", "vmcnt", "barrier2"]
            print("    cycles/tile (100 MHz ticks x clock ratio; shares matter): " +
                  ", ".join(f"{nm} {v / n:.0f}" for nm, v in zip(names, d[:7])) + f"  tiles {d[7]}")
    print(f"sum {tot*1e3:.1f} us")


if __name__ == "__main__":
    main()

"""Probe: fused C2f kernel (csrc/c2f_fused.hip) at the C2 bench size vs the four-launch path; MOY_C2F_DIAG=1 prints phase stamps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mo_yolo_amd import _lib as L, ops
B, H, W = int(os.environ.get("C2F_B", 288)), 152, 272
dt = torch.bfloat16
M = B * H * W
g = lambda *s, k=1.0: (torch.rand(*s, device="cuda") - 0.5) * k
x = g(M, 64).to(dt)
wp = dict(cv1=ops.pad_weight(g(64, 64, k=0.25), dt), m1=ops.pad_weight(g(32, 288, k=0.12), dt), m2=ops.pad_weight(g(32, 288, k=0.12), dt),
          cv2=ops.pad_weight(g(64, 96, k=0.2), dt))
arg = {k: (w, torch.rand(w.shape[0], device="cuda") + 0.5, g(w.shape[0], k=0.2)) for k, w in wp.items()}
out = torch.empty(M, 64, device="cuda", dtype=dt)
cat = torch.empty(M, 96, device="cuda", dtype=dt)
tmp = torch.empty(M, 32, device="cuda", dtype=dt)
out4 = torch.empty(M, 64, device="cuda", dtype=dt)
kw = lambda k: dict(scale=arg[k][1], shift=arg[k][2], act=L.ACT_SILU)
def fused():
    ops.c2f_fused(x, B, H, W, arg["cv1"], arg["m1"], arg["m2"], arg["cv2"], out=out)
def four():
    ops.gemm(x, wp["cv1"], 64, 64, out=cat[:, :64], **kw("cv1"))
    ops.gemm(cat[:, 32:64], wp["m1"], 32, 288, ksize=3, stride=1, geom=(B, H, W, H, W, 32), out=tmp, **kw("m1"))
    ops.gemm(tmp, wp["m2"], 32, 288, ksize=3, stride=1, geom=(B, H, W, H, W, 32), R=cat[:, 32:64], out=cat[:, 64:], **kw("m2"))
    ops.gemm(cat, wp["cv2"], 64, 96, out=out4, **kw("cv2"))
def timeit(f, n=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
if os.environ.get("MOY_C2F_DIAG") == "1":
    fused(); torch.cuda.synchronize()
    d = out.view(torch.int64).flatten()[:9].cpu().tolist()
    n = max(d[8], 1)
    names = ["cv1+x loads", "barrier", "m.cv1", "barrier", "m.cv2", "barrier", "cv2", "barrier+stores"]
    print("cycles/tile: " + ", ".join(f"{nm} {v / n:.0f}" for nm, v in zip(names, d[:8])) + f"  sum {sum(d[:8]) / n:.0f}  tiles {d[8]}")
else:
    tf, t4 = timeit(fused), timeit(four)
    print(f"C2f 64->[32|32]->64 B={B} {H}x{W}: fused {tf:.1f} us, four launches {t4:.1f} us; equal: "
          f"{float((out.float() - out4.float()).abs().max()):.4f} max abs diff")

"""Probe: phase stamps of the fused stem + layer-1 kernel (MOY_SL1_DIAG=1) and its time at the C2 bench size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mo_yolo_amd import ops
B, H, W = int(os.environ.get("SL1_B", 288)), 608, 1088
dt = torch.bfloat16
u8 = torch.randint(0, 256, (B, H, W, 3), dtype=torch.uint8, device="cuda")
w0 = ops.stem_weights_fused((torch.rand(32, 3, 3, 3, device="cuda") - 0.5) * 0.6, dt)
w1 = ops.pad_weight((torch.rand(64, 288, device="cuda") - 0.5) * 0.1, dt)
s0, h0, s1, h1 = (torch.rand(n, device="cuda") + 0.5 for n in (32, 32, 64, 64))
out = torch.empty(B * (H // 4) * (W // 4), 64, device="cuda", dtype=dt)
f = lambda: ops.stem_l1_fused(u8, w0, s0, h0, w1, s1, h1, dt, out=out)
f(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): f()
e1.record(); torch.cuda.synchronize()
print(f"stem+l1 fused B={B}: {e0.elapsed_time(e1) / 5 * 1e3:.1f} us")
if os.environ.get("MOY_SL1_DIAG") in ("1", "2", "3", "4"):
    d = out.view(torch.int64).flatten()[:8].cpu().tolist()
    n = max(d[7], 1)
    names = ["store_window", "barrier W", "stem", "barrier P", "layer1+epilogue", "barrier S", "stores"]
    print("cycles/tile: " + ", ".join(f"{nm} {v / n:.0f}" for nm, v in zip(names, d[:7])) + f"  tiles {d[7]}")

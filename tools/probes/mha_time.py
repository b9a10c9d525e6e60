"""Probe: the self-attention core (moy_mha_core) at the C2 bench shape (288 frames x 300 queries, 8 heads of 32): time per launch; --save / --cmp compare outputs of two builds."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mo_yolo_amd import ops
ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=288); ap.add_argument("--Lq", type=int, default=300); ap.add_argument("--dtype", default="bf16")
ap.add_argument("--save"); ap.add_argument("--cmp")
a = ap.parse_args()
dt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[a.dtype]
g = torch.Generator(device="cuda").manual_seed(1)
qkv = (torch.randn(a.B * a.Lq, 768, device="cuda", generator=g) * 1.5).to(dt)
f = lambda: ops.mha_core(qkv, a.B, a.Lq, 8)
y = f(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(5):
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 20 * 1e3)
print(f"mha_core {a.dtype} B={a.B} L={a.Lq}: " + " ".join(f"{t:.1f}" for t in ts) + " us per launch", flush=True)
if a.save: torch.save(y.cpu(), a.save)
if a.cmp:
    y0 = torch.load(a.cmp).float(); d = (y.float().cpu() - y0).abs()
    print(f"vs {a.cmp}: max |diff| {float(d.max()):.3e}, equal {float((d == 0).float().mean()):.4f}")

"""Probe: the two forms of the raw level-0 gather (csrc/msda_raw.hip) at the C2 bench shape -- time per launch and agreement.
Run twice: MOY_MR_MFMA=0 (vector-ALU tap sums) and MOY_MR_MFMA=1 (tap sums on the matrix cores); --save / --cmp <file> compare the outputs."""
import argparse, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mo_yolo_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=288)
ap.add_argument("--Lq", type=int, default=300)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--save")
ap.add_argument("--cmp")
ap.add_argument("--spread", type=float, default=1.0, help="scale of the sampling offsets")
a = ap.parse_args()
dt = {"bf16": torch.bfloat16, "f16": torch.float16}[a.dtype]
B, Lq, shapes = a.B, a.Lq, [(76, 136), (38, 68), (19, 34)]
g = torch.Generator(device="cuda").manual_seed(1)
H0, W0 = shapes[0]
S1 = sum(h * w for h, w in shapes[1:])
x = torch.randn(B * H0 * W0, 128, device="cuda", generator=g).to(dt)
wc = (torch.randn(256, 128, device="cuda", generator=g) / math.sqrt(128)).to(dt)
bc = torch.randn(256, device="cuda", generator=g) * 0.5
planes = torch.randn(8, B * S1, 32, device="cuda", generator=g).to(dt)
offaw = torch.cat([torch.randn(B * Lq, 192, device="cuda", generator=g) * a.spread, torch.randn(B * Lq, 96, device="cuda", generator=g)], 1).contiguous()
ref = torch.rand(B * Lq, 4, device="cuda", generator=g)
ref[:, 2:] = ref[:, 2:] * 0.3 + 0.02
PACKED = os.environ.get("WC_PACKED", "1") != "0"
wcp = ops.pack_mfma_a(wc) if PACKED else wc
f = lambda: ops.msda_raw0(x, wcp, bc, planes, B, shapes, offaw, ref, Lq, packed=PACKED)
y = f(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(5):
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 10 * 1e3)
print(f"msda_raw0 MOY_MR_MFMA={os.environ.get('MOY_MR_MFMA', '(default)')} wc {'fragment order' if PACKED else 'row-major'} {a.dtype} B={B} Lq={Lq}: " + " ".join(f"{t:.1f}" for t in ts) + " us per launch", flush=True)
if a.save:
    torch.save(y.cpu(), a.save)
if a.cmp:
    y0 = torch.load(a.cmp).float()
    d = (y.float().cpu() - y0).abs()
    print(f"vs {a.cmp}: max |diff| {float(d.max()):.3e}, mean {float(d.mean()):.3e}, max |y| {float(y0.abs().max()):.3f}, equal {float((d == 0).float().mean()):.4f}")

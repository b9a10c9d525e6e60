"""Probe: time moy_gemm in split fp16 precision (MOY_F32X3) against the exact fp32 kernel on the shapes that dominate the fp32 engines."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mo_yolo_amd import ops, _lib as L
dev = "cuda"
def t(f, n=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for name, M, N, K in (("value projection", 1302336, 1536, 256), ("1x1 conv", 992256, 256, 128), ("linear M=28800", 28800, 768, 256)):
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) / K ** 0.5
    w32, w3 = ops.pad_weight(w, torch.float32), ops.split_weight(w)
    out = torch.empty(M, N, device=dev)
    ms3 = t(lambda: ops.gemm(x, w3, N, K, out=out, split_f16=True))
    ms1 = t(lambda: ops.gemm(x, w32, N, K, out=out)) if os.environ.get("X3_ONLY") != "1" else float("nan")
    fl = 2 * M * N * K
    print(f"{name}: x3 {ms3:.3f} ms = {fl / ms3 / 1e9:.0f} TF-eq   fp32 {ms1:.3f} ms = {fl / ms1 / 1e9:.0f} TF")
B, H, W, C = 96, 76, 136, 128
x = torch.randn(B * H * W, C, device=dev)
w = torch.randn(C, 9 * C, device=dev) / (9 * C) ** 0.5
w3 = ops.split_weight(w)
sc, sh = torch.ones(C, device=dev), torch.zeros(C, device=dev)
out = torch.empty(B * H * W, C, device=dev)
ms3 = t(lambda: ops.gemm(x, w3, C, 9 * C, ksize=3, stride=1, geom=(B, H, W, H, W, C), scale=sc, shift=sh, act=L.ACT_SILU, out=out, split_f16=True))
print(f"conv3x3 C128: x3 {ms3:.3f} ms = {2 * B * H * W * C * 9 * C / ms3 / 1e9:.0f} TF-eq")

"""Probe: what does this device reach for a pure write stream, a pure read stream and a copy?  (the value projection writes 12 GB per launch)"""
import torch
def t(f, n=5):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3
N = 6 * 1024 ** 3            # elements (bf16): 12 GiB
x = torch.empty(N, dtype=torch.bfloat16, device="cuda")
y = torch.empty(N, dtype=torch.bfloat16, device="cuda")
print(f"fill  12 GiB: {N * 2 / t(lambda: x.fill_(1.0)) / 1e12:.2f} TB/s written")
print(f"copy  12 GiB: {2 * N * 2 / t(lambda: y.copy_(x)) / 1e12:.2f} TB/s (read + write)")
s = x.view(torch.int32)
print(f"read  12 GiB (sum): {N * 2 / t(lambda: s.sum()) / 1e12:.2f} TB/s read")

#!/usr/bin/env python3
"""One profile line for the operator's backward: ms_deform_attn_forward + ms_deform_attn_backward at the C2
decoder shape a training step of the upstream model would see (N frames x 300 queries, 8 heads x 32 channels, 3 levels x 4 points over
the 1088x608 pyramid), fp32, through `MSDeformAttnFunction`.  Run under `rocprofv3 --kernel-trace --stats`; prints us per call and the
algorithmic bytes / flops of the backward (grad_value atomics: N*Lq*M*L*P*4 taps x D channels x 4 B read-modify-write)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mo_yolo_amd.modules import MSDeformAttnFunction

N, M, D, Lq, P = int(os.environ.get("MB_N", 8)), 8, 32, 300, 4
shapes_l = [(76, 136), (38, 68), (19, 34)]
S = sum(h * w for h, w in shapes_l)
g = torch.Generator().manual_seed(1)
value = (torch.rand(N, S, M, D, generator=g) - 0.5).cuda().requires_grad_(True)
loc = (torch.rand(N, Lq, M, 3, P, 2, generator=g) * 1.1 - 0.05).cuda().requires_grad_(True)
aw = torch.rand(N, Lq, M, 3, P, generator=g)
aw = (aw / aw.sum((-1, -2), keepdim=True)).cuda().requires_grad_(True)
shapes = torch.tensor(shapes_l, dtype=torch.int64).cuda()
lsi = torch.cat((shapes.new_zeros(1), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]))
go = (torch.rand(N, Lq, M * D, generator=g) - 0.5).cuda()


def step():
    for t in (value, loc, aw):
        t.grad = None
    y = MSDeformAttnFunction.apply(value, shapes, lsi, loc, aw, 64)
    y.backward(go)


for _ in range(3):
    step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 20
e0.record()
for _ in range(reps):
    step()
e1.record()
torch.cuda.synchronize()
taps = N * Lq * M * 3 * P * 4
print(f"msda fwd+bwd N={N} Lq={Lq} S={S}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us per fwd+bwd; backward algorithmic: "
      f"{taps * D * 4 * 2 / 1e6:.1f} MB of grad_value read-modify-write (atomics), {N * S * M * D * 4 / 1e6:.1f} MB cleared, "
      f"{taps * D * 2 * 3 / 1e9:.2f} GFLOP")

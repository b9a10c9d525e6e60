"""Probe: C4 bf16 engine at bench batch vs small batch vs fp32 (separates dtype effects from kernel-selection effects)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mo_yolo_amd.engine import TrackEngine
from mo_yolo_amd.fixtures import fixture
from mo_yolo_amd.parity import engine_pair_stats
from mo_yolo_amd.synth import SyntheticSequence
name = sys.argv[1] if len(sys.argv) > 1 else "c4"
Bbig = int(sys.argv[2]) if len(sys.argv) > 2 else 64
cfg, arch, sd = fixture(name)
seq = SyntheticSequence(0, cfg["H"], cfg["W"], cfg["style"])
fr = torch.from_numpy(seq.frames(0, Bbig)).to("cuda")
def run(dt, B, conv_ws=None):
    e = TrackEngine(arch, sd, cfg["H"], cfg["W"], batch=B, dtype=dt)
    outs = []
    for t0 in range(0, 4, B):
        o = e.forward(fr[t0:t0 + B] if B < Bbig else fr)
        torch.cuda.synchronize()
        outs.append({k: v[:4].cpu().clone() for k, v in o.items() if hasattr(v, "shape") and v.shape[:1] == (B,)})
        if B >= 4:
            break
    return {k: torch.cat([o[k] for o in outs])[:4] for k in outs[0]}
f32 = run(torch.float32, 2)
b_small = run(torch.bfloat16, 2)
b_big = run(torch.bfloat16, Bbig)
for tag, a, b in (("bf16 big vs bf16 small", b_big, b_small), ("bf16 small vs f32", b_small, f32), ("bf16 big vs f32", b_big, f32)):
    st = engine_pair_stats(a, b, arch.nq)
    print(tag, {k: st[k] for k in ("topk_overlap", "box_max_err_matched", "score_max_err_matched", "hs_max_err_matched", "births_flipped", "active_rows_reference", "active_rows")})
# signed logit difference on matched rows (bias?)
import numpy as np
for tag, a in (("big", b_big), ("small", b_small)):
    d = []
    for b in range(4):
        pos = {int(t): i for i, t in enumerate(f32["topk_ind"][b].tolist())}
        for i, t in enumerate(a["topk_ind"][b].tolist()):
            j = pos.get(int(t))
            if j is not None:
                d.append(float(a["logits"][b, i, 0] - f32["logits"][b, j, 0]))
    d = np.array(d)
    print(f"bf16 {tag} - f32 logits on matched rows: mean {d.mean():+.4f} std {d.std():.4f} max|.| {np.abs(d).max():.4f}")

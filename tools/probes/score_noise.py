"""Probe: how far apart are two correct fp32 evaluations of the encoder scores (engine on the GPU vs the CPU oracle)?  Sets the
separation the fixtures need between adjacent top-k scores (tests/golden/make_golden.py: separate_topk)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from mo_yolo_amd.engine import TrackEngine
from mo_yolo_amd.fixtures import fixture
from mo_yolo_amd.synth import SyntheticSequence, to_network_input
from oracle import track_oracle as O

name = sys.argv[1] if len(sys.argv) > 1 else "c2"
cfg, arch, sd = fixture(name)
B = 2
fr = SyntheticSequence(0, cfg["H"], cfg["W"], cfg["style"]).frames(0, B)
eng = TrackEngine(arch, sd, cfg["H"], cfg["W"], batch=B, dtype=torch.float32)
out = eng.forward(torch.from_numpy(fr).to("cuda"))
torch.cuda.synchronize()
S = eng.S
got = eng.scores_all.view(B, S, -1).max(-1).values.cpu().double()
with torch.no_grad():
    r = O.forward(to_network_input(fr), sd, arch)
want = r["enc_scores_all"].max(-1).values.double()
valid = r["valid"][0, :, 0]
for b in range(B):
    d = (got[b] - want[b]).abs()
    top = torch.topk(want[b], arch.nq).indices
    srt = torch.sort(want[b], descending=True).values[:arch.nq + 1]
    gaps = srt[:-1] - srt[1:]
    print(f"frame {b}: |score| of the top-{arch.nq}: {float(srt[-1]):.4f}..{float(srt[0]):.4f}; engine-vs-oracle abs diff: valid tokens max "
          f"{float(d[valid].max()):.3e} median {float(d[valid].median()):.3e}; top-k tokens max {float(d[top].max()):.3e} median "
          f"{float(d[top].median()):.3e}; adjacent gaps: min {float(gaps.min()):.3e} mean {float(gaps.mean()):.3e}; top-k equal: "
          f"{bool(torch.equal(out['topk_ind'][b].cpu().long(), r['topk_ind'][b]))}")

#!/usr/bin/env python3
"""The sequence of tests/test_gpu_engine.py::test_engine_folded_input_proj_plan_vs_classic_plan repeated in one process: fresh
classic + folded engines per repetition and dtype, one forward each, value planes / scores compared; where they differ by more
than the rounding budget, the plane, token range and level of the outliers are printed (a localised corruption names its kernel)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from mo_yolo_amd.engine import TrackEngine  # noqa: E402
from mo_yolo_amd.fixtures import fixture  # noqa: E402
from mo_yolo_amd.synth import SyntheticSequence  # noqa: E402

cfg, arch, sd = fixture("c2")
H, W, B = cfg["H"], cfg["W"], int(os.environ.get("FVC_B", 104))
seq = SyntheticSequence(0, H, W, cfg["style"])
fr = torch.from_numpy(np.concatenate([seq.frames(t, 1) for t in range(B)])).cuda()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
nbad = 0
for rep in range(reps):
    for dt in (torch.bfloat16, torch.float16):
        os.environ["MOY_FOLD_PROJ"] = "0"
        classic = TrackEngine(arch, sd, H, W, batch=B, dtype=dt)
        os.environ["MOY_FOLD_PROJ"] = "1"
        folded = TrackEngine(arch, sd, H, W, batch=B, dtype=dt)
        classic.forward(fr)
        folded.forward(fr)
        torch.cuda.synchronize()
        S = folded.S
        vc, vf = classic.value_planes.float().view(-1, B, S, 32), folded.value_planes.float().view(-1, B, S, 32)
        eps = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
        d = (vc - vf).abs()
        lim = 16 * eps * float(vc.abs().max())
        # backbone outputs must be IDENTICAL (same kernels, same input)
        lay = [i for i, v in classic.layer_views.items() if v is not None and i not in classic.virtual_layers
               and not torch.equal(v.tensor(), folded.layer_views[i].tensor())]
        msg = f"rep {rep} {dt}: max value diff {float(d.max()):.4f} (limit {lim:.4f}); backbone layers that differ: {lay}"
        if float(d.max()) > lim or lay:
            nbad += 1
            bad = (d > lim).nonzero()
            if len(bad):
                pl, bb, tok = bad[:, 0], bad[:, 1], bad[:, 2]
                msg += (f"\\n   {len(bad)} outliers: planes {sorted(set(pl.tolist()))[:12]} frames {sorted(set(bb.tolist()))[:12]} tokens "
                        f"{int(tok.min())}..{int(tok.max())} (levels start at 0, {76 * 136}, {76 * 136 + 38 * 68})")
                # which engine is off: compare both with an fp32 product of the folded engine's own inputs is costly; report the values
                i0 = bad[0].tolist()
                msg += f"\\n   first outlier {i0}: classic {float(vc[tuple(i0)]):.4f} folded {float(vf[tuple(i0)]):.4f}"
        print(msg, flush=True)
        del classic, folded
print("bad:", nbad)

"""Fixture configurations of the path (BASELINE.json configs C2 / C4 + the small parity cases): seeded synthetic weights under
the reference's parameter names (weights.py) plus the calibration overlay found with the imported reference
(tests/golden/make_golden.py -> data/fixture_calib.npz).  Lives in the package so that bench.py, the driver entry points and
the tests share one definition and the product never imports from tests/."""
from __future__ import annotations

import functools

from .config import build_arch
from .weights import apply_calibration, make_fixture_state_dict

CONFIGS = {
    # name: depth, width, nc, H, W, nq, weight seed, stream style, golden frames
    "tiny": dict(depth=0.33, width=0.25, nc=1, H=96, W=160, nq=50, seed=0, style="mot17", frames=3),
    "tiny3": dict(depth=0.33, width=0.25, nc=3, H=64, W=96, nq=20, seed=3, style="mot17", frames=4),
    "c2": dict(depth=0.33, width=0.50, nc=1, H=608, W=1088, nq=300, seed=0, style="mot17", frames=8),
    "c4": dict(depth=0.33, width=0.50, nc=1, H=1088, W=1920, nq=500, seed=0, style="dance", frames=2),
    # yolo_track.yaml AS SHIPPED (depth 1.0 / width 1.0, the scale the reference's own entry script uses: start_train.py:11,
    # cfg/models/v8/yolo_track.yaml:11-12; 46 M parameters) at a small resolution: the widths no specialised kernel covers
    "full": dict(depth=1.0, width=1.0, nc=1, H=128, W=192, nq=60, seed=0, style="mot17", frames=2),
    # the same 46 M-parameter weights at the headline resolution and query count: a TIMING configuration (bench.py --config full);
    # its calibration overlay is "full"'s (BatchNorm statistics and score heads found at 128 x 192), no golden of its own
    "full_c2": dict(depth=1.0, width=1.0, nc=1, H=608, W=1088, nq=300, seed=0, style="mot17", frames=0, calib="full"),
}


@functools.lru_cache(maxsize=None)
def fixture(name: str):
    """(cfg dict, arch, state_dict) of fixture config `name`: weights regenerated from the seed, calibration overlaid."""
    cfg = dict(CONFIGS[name], name=name)
    arch = build_arch(cfg["depth"], cfg["width"], cfg["nc"], cfg["nq"])
    sd = make_fixture_state_dict(arch, cfg["seed"])
    apply_calibration(sd, cfg.get("calib", name))
    return cfg, arch, sd

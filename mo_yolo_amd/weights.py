"""Synthetic, de-degenerated fixture weights + neutral weight-file I/O.

The reference ships no weights and its default init is degenerate for parity work
(zero sampling_offsets/attention_weights, zero bbox-head last layers, all
300 scores equal).  This module generates a *seeded* state_dict with the reference key names
(config.param_shapes) from integer->float32 uniform draws only (numpy PCG64 `random()`), so the
same bytes are produced in the build container (where the goldens are made) and on the GPU box.

Small data-dependent calibration vectors (encoder bias trick, last score head) are found once
with the imported reference by tests/golden/make_golden.py and stored in
mo_yolo_amd/data/fixture_calib.npz; `apply_calibration` overlays them.
"""
from __future__ import annotations

import hashlib
import math
import os
from collections import OrderedDict

import numpy as np
import torch

from .config import TrackArch, param_shapes

_DATA = os.path.join(os.path.dirname(__file__), "data")
DETECT_CLS_BIAS = 8.9     # Detect class-logit offset of the C1 fixture (a few % of the anchors above conf 0.25)


def _u(rng, shape, a):
    """float32 U(-a, a) from exact integer->float conversion (platform independent)."""
    x = rng.random(size=shape, dtype=np.float32)
    return ((x * 2.0 - 1.0) * np.float32(a)).astype(np.float32)


def _ur(rng, shape, lo, hi):
    x = rng.random(size=shape, dtype=np.float32)
    return (np.float32(lo) + x * np.float32(hi - lo)).astype(np.float32)


def _offset_grid_bias(nh, nl, npnt):
    # unit directions on the max-norm circle, scaled by (point index + 1): the usual
    # deformable-attention offset prior (transformer.py:218-230 documents the same prior).
    th = np.arange(nh, dtype=np.float32) * np.float32(2.0 * math.pi / nh)
    g = np.stack([np.cos(th), np.sin(th)], -1).astype(np.float32)
    g = g / np.abs(g).max(-1, keepdims=True)
    g = np.tile(g[:, None, None, :], (1, nl, npnt, 1))
    for p in range(npnt):
        g[:, :, p, :] *= p + 1
    return g.reshape(-1).astype(np.float32)


def make_fixture_state_dict(arch: TrackArch, seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    rng = np.random.Generator(np.random.PCG64(seed))
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    shapes = param_shapes(arch)
    s3 = math.sqrt(3.0)
    for k, shp in shapes.items():
        leaf = k.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            v = np.zeros((), np.int64)
        elif leaf == "running_mean":
            v = _u(rng, shp, 0.17)
        elif leaf == "running_var":
            v = _ur(rng, shp, 0.8, 1.2)
        elif k.endswith("dfl.conv.weight"):                   # DFL: fixed arange(16) (block.py:24-26)
            v = np.arange(16, dtype=np.float32).reshape(shp)
        elif len(shp) == 4:                                   # conv weight: He-uniform
            fan_in = shp[1] * shp[2] * shp[3]
            v = _u(rng, shp, math.sqrt(6.0 / fan_in))
            if (".cv2." in k or ".cv3." in k) and k.endswith(".2.weight"):
                v = v * np.float32(20.0)                      # Detect output convs: spread the DFL / class logits
        elif ".bn." in k and arch.head_kind == "track":
            # Conv blocks of the tracking fixtures (round 3): gamma in [0.4, 0.6], beta in [0.3, 0.7].  With the BatchNorm
            # statistics calibrated (tests/golden/make_golden.py:calibrate_bn) the pre-activation is z ~ N(beta, gamma^2); a
            # RANDOM network then amplifies any relative perturbation by sqrt(E[silu'(z)^2] gamma^2 / Var silu(z)) per layer
            # (Gaussian Poincare inequality: >= 1, equality only for a linear activation): 1.10 for gamma 1 / beta 0, i.e. the
            # 0.3 % of one bf16 rounding grows to 8 % over the 46 convolutions of the longest path (measured:
            # profiles/r03_seam_bias_c2_bf16_bn_gamma1.json) -- a property of random weights, not of 16-bit kernels and not of
            # trained networks.  This range gives 1.02 per layer.  Same number of draws as before: every other weight is unchanged.
            v = _ur(rng, shp, 0.4, 0.6) if leaf == "weight" else _ur(rng, shp, 0.3, 0.7)
        elif ".bn." in k or ".input_proj." in k or "norm" in k or k.endswith("enc_output.1.weight") \
                or k.endswith("enc_output.1.bias"):
            v = _ur(rng, shp, 0.8, 1.2) if leaf == "weight" else _u(rng, shp, 0.17)
        elif leaf == "in_proj_weight":
            v = _u(rng, shp, s3 / 16.0)
        elif leaf == "in_proj_bias":
            v = _u(rng, shp, 0.05)
        elif "sampling_offsets.weight" in k:
            v = _u(rng, shp, 0.02 * s3)
        elif "sampling_offsets.bias" in k:
            v = _offset_grid_bias(arch.nh, arch.nl, arch.ndp) + _u(rng, shp, 0.1)
        elif "attention_weights.weight" in k:
            v = _u(rng, shp, 0.05 * s3)
        elif "attention_weights.bias" in k:
            v = _u(rng, shp, 0.5 * s3)
        elif leaf == "weight" and len(shp) == 2:              # Linear / Embedding
            v = _u(rng, shp, s3 / math.sqrt(shp[1]))
            if ("bbox_head" in k and k.endswith("layers.2.weight")):
                v = v * np.float32(0.05)
        elif leaf == "bias" and ".cv3." in k:               # Detect class bias: keep most anchors below conf 0.25
            v = _u(rng, shp, 0.05) - np.float32(DETECT_CLS_BIAS)
        elif leaf == "bias":
            v = _u(rng, shp, 0.05)
            if ("bbox_head" in k and k.endswith("layers.2.bias")):
                v = v * np.float32(0.05)
        else:
            raise KeyError(k)
        sd[k] = torch.from_numpy(np.ascontiguousarray(v))
    # Masked-token guard: every masked token has the feature
    # LN(enc_output.bias); anti-align that bias with the score direction so the constant
    # masked-token score ranks below the valid tokens and top-k never selects a +inf anchor.
    if arch.head_kind == "track":
        d = f"model.{len(arch.layers)}.decoder"
        w = sd[d + ".enc_score_head.weight"]
        sd[d + ".enc_output.0.bias"] = (-0.02 * w.mean(0)).contiguous()
    return sd


def state_dict_digest(sd) -> str:
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(v.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def calib_path():
    return os.path.join(_DATA, "fixture_calib.npz")


def apply_calibration(sd, name: str, strict: bool = True):
    """Overlay the stored calibration vectors for fixture config `name` (in place)."""
    p = calib_path()
    if not os.path.exists(p):
        if strict:
            raise FileNotFoundError(p)
        return sd
    z = np.load(p)
    pref = name + "/"
    hit = False
    for key in z.files:
        if key.startswith(pref):
            sd[key[len(pref):]] = torch.from_numpy(z[key].copy())
            hit = True
    if strict and not hit:
        raise KeyError(f"no calibration stored for fixture config '{name}'")
    return sd


def save_weights(sd, path):
    """Neutral weight file: npz of the state_dict under the reference key names (fp32)."""
    np.savez(path, **{k: v.detach().cpu().numpy() for k, v in sd.items()})


def load_weights(path) -> "OrderedDict[str, torch.Tensor]":
    z = np.load(path)
    return OrderedDict((k, torch.from_numpy(z[k])) for k in z.files)

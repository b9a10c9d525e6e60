"""Shared helpers for the tests: fixture configs, weights, goldens."""
import ast
import functools
import os

import numpy as np
import torch

from mo_yolo_amd.config import build_arch
from mo_yolo_amd.synth import SyntheticSequence, to_network_input
from mo_yolo_amd.weights import apply_calibration, make_fixture_state_dict, state_dict_digest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@functools.lru_cache(maxsize=None)
def golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


@functools.lru_cache(maxsize=None)
def fixture(name):
    """(cfg dict, arch, state_dict) for a golden config; weights are re-generated from the seed and
    checked against the digest stored when the goldens were made."""
    g = golden(name)
    cfg = ast.literal_eval(str(g["cfg"]))
    arch = build_arch(cfg["depth"], cfg["width"], cfg["nc"], cfg["nq"])
    sd = make_fixture_state_dict(arch, cfg["seed"])
    apply_calibration(sd, name)
    assert state_dict_digest(sd) == str(g["weights_sha256"]), "fixture weights drifted from the goldens"
    return cfg, arch, sd


def frames_u8(cfg, t0, n, seq_id=0):
    return SyntheticSequence(seq_id, cfg["H"], cfg["W"], cfg["style"]).frames(t0, n)


def net_input(cfg, t0, n, seq_id=0):
    return to_network_input(frames_u8(cfg, t0, n, seq_id))


def gold_tensor(g, key):
    """Full tensor if stored, else (idx, val, shape) sample triple."""
    if key in g:
        return torch.from_numpy(g[key]), None
    return torch.from_numpy(g[key + ".val"]), torch.from_numpy(g[key + ".idx"])


def check_close(got, g, key, atol, rtol=0.0, what=""):
    want, idx = gold_tensor(g, key)
    got = got.detach().float().cpu()
    if idx is not None:
        assert tuple(got.shape) == tuple(g[key + ".shape"]), (key, got.shape, g[key + ".shape"])
        got = got.reshape(-1)[idx]
    else:
        assert tuple(got.shape) == tuple(want.shape), (key, got.shape, want.shape)
    err = (got - want).abs()
    tol = atol + rtol * want.abs()
    bad = err > tol
    assert not bool(bad.any()), f"{what}{key}: max err {err.max().item():.3e} (tol {atol}+{rtol}*|x|), {int(bad.sum())} bad"
    return float(err.max())

"""Shared helpers for the tests: fixture configs, weights, goldens."""
import ast
import functools
import os

import numpy as np
import torch

from mo_yolo_amd import fixtures as _fx
from mo_yolo_amd.synth import SyntheticSequence, to_network_input
from mo_yolo_amd.weights import state_dict_digest

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
    cfg, arch, sd = _fx.fixture(name)
    gcfg = ast.literal_eval(str(g["cfg"]))
    assert all(cfg[k] == v for k, v in gcfg.items() if k in cfg), "fixture config drifted from the goldens"
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


def hota_of_tracks(cfg, per_frame_rows, per_frame_ids, seq_id=0):
    """HOTA (the reference evaluator's algorithm, oracle/hota_oracle.py) of per-frame track rows
    (xyxy pixels) + ids against the synthetic sequence's ground truth.  Data layout as the validator
    builds it (val.py:418-432): ids are [n,1] int arrays re-indexed to 0..N-1, similarity = box IoU."""
    from oracle import hota_oracle as H
    seq = SyntheticSequence(seq_id, cfg["H"], cfg["W"], cfg["style"])
    T = len(per_frame_rows)
    gts = [seq.boxes(t) for t in range(T)]
    ug = np.unique(np.concatenate([g[1] for g in gts]))
    ut = np.unique(np.concatenate([np.asarray(i).reshape(-1) for i in per_frame_ids] + [np.zeros(0, np.int64)]))
    gi, ti, sims = [], [], []
    for t in range(T):
        gb, gid = gts[t]
        tb = np.asarray(per_frame_rows[t], dtype=np.float64).reshape(-1, 4)
        gi.append(np.searchsorted(ug, gid).reshape(-1, 1))
        ti.append(np.searchsorted(ut, np.asarray(per_frame_ids[t]).reshape(-1)).reshape(-1, 1))
        ix1 = np.maximum(gb[:, None, 0], tb[None, :, 0]); iy1 = np.maximum(gb[:, None, 1], tb[None, :, 1])
        ix2 = np.minimum(gb[:, None, 2], tb[None, :, 2]); iy2 = np.minimum(gb[:, None, 3], tb[None, :, 3])
        inter = np.clip(ix2 - ix1, 0, None) * np.clip(iy2 - iy1, 0, None)
        ua = (gb[:, 2] - gb[:, 0]) * (gb[:, 3] - gb[:, 1]); ub = (tb[:, 2] - tb[:, 0]) * (tb[:, 3] - tb[:, 1])
        sims.append(inter / np.maximum(ua[:, None] + ub[None, :] - inter, 1e-9))
    return H.eval_sequence(gi, ti, sims, len(ug), max(1, len(ut)))

"""HOTA as the reference's validator computes it -- TEST INFRASTRUCTURE (see oracle/__init__.py).

Restates `HOTA.eval_sequence` of ultralytics/utils/hota.py:24-164 (a TrackEval port the fork
patched with try/except fallbacks) for the input layout the validator builds
(models/MOTRtrack/val.py:418-432, 298-308): per timestep `gt_ids[t]` int [n,1], `tracker_ids[t]`
int [K,1], `similarity_scores[t]` float [n,K] (IoU of pixel xyxy boxes).  The fallbacks change the
result, so they are part of the parity target:
  * the potential-match accumulation indexes by detection POSITION whenever numpy cannot broadcast
    the identity index (hota.py:63-71), i.e. for every frame with more than one tracker row;
  * tracker ids are shifted IN PLACE (minus the frame's minimum on the first frame that has an
    id >= 1, then minus 1 on every frame, hota.py:81-88) and those shifted ids are what the second
    pass and the association counts use (-1 wraps to the last column);
  * the alignment scores of frame t are read from the first K COLUMNS (hota.py:116-117).
Pinned against reference outputs in tests/golden/hota.npz.
"""
from __future__ import annotations

import numpy as np
from scipy.optimize import linear_sum_assignment

ALPHAS = np.arange(0.05, 0.99, 0.05)      # hota.py:16
EPS = np.finfo("float").eps


def _final_fields(res):                    # hota.py:217-231
    tp, fn, fp = res["HOTA_TP"], res["HOTA_FN"], res["HOTA_FP"]
    res["DetRe"] = tp / np.maximum(1, tp + fn)
    res["DetPr"] = tp / np.maximum(1, tp + fp)
    res["DetA"] = tp / np.maximum(1, tp + fn + fp)
    res["HOTA"] = np.sqrt(res["DetA"] * res["AssA"])
    res["OWTA"] = np.sqrt(res["DetRe"] * res["AssA"])
    res["HOTA(0)"], res["LocA(0)"] = res["HOTA"][0], res["LocA"][0]
    res["HOTALocA(0)"] = res["HOTA(0)"] * res["LocA(0)"]
    return res


def eval_sequence(gt_ids, tracker_ids, sims, num_gt_ids, num_tracker_ids):
    nA = len(ALPHAS)
    res = {k: np.zeros(nA) for k in ("HOTA", "DetA", "AssA", "DetRe", "DetPr", "AssRe", "AssPr", "LocA", "OWTA",
                                     "HOTA_TP", "HOTA_FN", "HOTA_FP")}
    n_trk = sum(len(t) for t in tracker_ids)
    n_gt = sum(len(g) for g in gt_ids)
    if n_trk == 0 or n_gt == 0:             # hota.py:36-46
        res["HOTA_FN" if n_trk == 0 else "HOTA_FP"] = float(n_gt if n_trk == 0 else n_trk) * np.ones(nA)
        res["LocA"] = np.ones(nA)
        res.update({"HOTA(0)": 0, "LocA(0)": 1.0, "HOTALocA(0)": 0})
        return res
    trk = [np.array(t, dtype=np.int64).reshape(-1, 1) for t in tracker_ids]     # shifted in place below
    gts = [np.array(g, dtype=np.int64).reshape(-1, 1) for g in gt_ids]
    pot = np.zeros((num_gt_ids, num_tracker_ids))
    g_cnt = np.zeros((num_gt_ids, 1))
    t_cnt = np.zeros((1, num_tracker_ids))
    # ---- pass 1: global association statistics
    for g2, tr, sim in zip(gts, trk, sims):
        if len(g2) < 1:
            continue
        g = g2[:, 0]
        den = sim.sum(0)[None, :] + sim.sum(1)[:, None] - sim
        jac = np.where(den > EPS, sim / np.where(den > EPS, den, 1.0), 0.0)
        try:
            pot[g[:, None], tr[None, :]] += jac          # identity index: only broadcastable when K == 1
        except (ValueError, IndexError):
            pot[:len(g), :len(tr)] += jac                # the branch the shipped code takes: by position
        if len(g_cnt) <= g.max():
            g -= 1
        try:
            g_cnt[g] += 1
        except IndexError:
            pass
        try:
            if len(t_cnt) <= tr.max():
                tr -= tr.min()
        except ValueError:
            pass
        tr -= 1
        try:
            t_cnt[tr] += 1
        except IndexError:
            t_cnt = t_cnt[0]
            t_cnt[tr] += 1
    align = pot / (g_cnt + t_cnt - pot)
    matches = [np.zeros_like(pot) for _ in range(nA)]
    # ---- pass 2: per-frame Hungarian matching on alignment x similarity
    for g2, tr, sim in zip(gts, trk, sims):
        n, K = len(g2), len(tr)
        if n == 0:
            res["HOTA_FP"] += K
            continue
        if K == 0:
            res["HOTA_FN"] += n
            continue
        score = np.squeeze(align[g2[:, None], 0:K]) * sim          # hota.py:118-119: squeeze drops every unit axis
        try:                                                       # hota.py:122-131
            rows, cols = linear_sum_assignment(-score)
        except ValueError:
            score = np.squeeze(score)
            try:
                rows, cols = linear_sum_assignment(-score)
            except ValueError:
                rows, cols = linear_sum_assignment(-score[0, :, :])
        for a, alpha in enumerate(ALPHAS):
            try:                                                   # statement order of hota.py:134-149
                ok = sim[rows, cols] >= alpha - EPS
                r_, c_ = rows[ok], cols[ok]
                m = len(r_)
                res["HOTA_TP"][a] += m
                res["HOTA_FN"][a] += n - m
                res["HOTA_FP"][a] += K - m
                if m > 0:
                    res["LocA"][a] += sum(sim[r_, c_])
                    matches[a][g2[r_], tr[c_]] += 1
            except IndexError:
                res["HOTA_FN"][a] += n
                res["HOTA_FP"][a] += K
    for a in range(nA):
        mc = matches[a]
        tp = np.maximum(1, res["HOTA_TP"][a])
        res["AssA"][a] = np.sum(mc * (mc / np.maximum(1, g_cnt + t_cnt - mc))) / tp
        res["AssRe"][a] = np.sum(mc * (mc / np.maximum(1, g_cnt))) / tp
        res["AssPr"][a] = np.sum(mc * (mc / np.maximum(1, t_cnt))) / tp
    res["LocA"] = np.maximum(1e-10, res["LocA"]) / np.maximum(1e-10, res["HOTA_TP"])
    return _final_fields(res)


def box_ious_xyxy(b1, b2):
    """`TrackValidator._calculate_box_ious(..., box_format='x0y0x1y1')`, models/MOTRtrack/val.py:517-553."""
    b1 = np.asarray(b1, dtype=np.float32).reshape(-1, 4); b2 = np.asarray(b2, dtype=np.float32).reshape(-1, 4)
    mn = np.minimum(b1[:, None, :], b2[None, :, :]); mx = np.maximum(b1[:, None, :], b2[None, :, :])
    inter = np.maximum(mn[..., 2] - mx[..., 0], 0) * np.maximum(mn[..., 3] - mx[..., 1], 0)
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1]); a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    union = a1[:, None] + a2[None, :] - inter
    inter[a1 <= EPS, :] = 0
    inter[:, a2 <= EPS] = 0
    inter[union <= EPS] = 0
    union[union <= EPS] = 1
    return inter / union

"""Validation counterpart of the tracking path: sequence-wise HOTA with the
similarity matrices computed on the device and the assignment problem solved on the host.

Reference: `TrackValidator.__call__` (ultralytics/models/MOTRtrack/val.py:185-507: per frame keep the
rows with `obj_idxes >= 0` :383-386, accumulate gt / tracker ids and boxes :418-432, at the end of a
sequence build IoU similarities :298-308 with `_calculate_box_ious` :517-553 and call
`HOTA().eval_sequence` :310) and `HOTA` (ultralytics/utils/hota.py:8-231, a TrackEval port).

Two evaluators behind the reference's `eval_sequence(data)` / `combine_sequences(all_res)` API:

* `HOTA(compat=True)` (default) reproduces what the reference's patched file computes, including the
  branches its try/except blocks take for the validator's `[n, 1]` id layout (hota.py:63-97,116-117):
  that is the parity target, pinned by tests/golden/hota.npz (outputs of the reference itself).
* `HOTA(compat=False)` is the published HOTA definition (Luiten et al., IJCV 2021; TrackEval): ids
  index the association matrices by identity.  Use it for numbers comparable with the literature.

The reference validator hands `pred_boxes` (normalised cx,cy,w,h) to an x0y0x1y1 IoU against pixel
ground truth (val.py:418-432), which makes its similarities meaningless; SURVEY §3.4 fixes the layout
as "IoU in pixel xyxy" and this validator follows that (deliberate deviation, results of the
reference's own call are discarded anyway: val.py:310-316).
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Sequence

import numpy as np
import torch
from scipy.optimize import linear_sum_assignment

from . import ops

_EPS = np.finfo("float").eps


class HOTA:
    """`ultralytics/utils/hota.py:8-231`: same fields, same `eval_sequence(data)` input dictionary
    (`num_gt_ids`, `num_tracker_ids`, `gt_ids`, `tracker_ids`, `similarity_scores`; hota.py:24-34)."""

    array_labels = np.arange(0.05, 0.99, 0.05)                                  # hota.py:16
    integer_array_fields = ["HOTA_TP", "HOTA_FN", "HOTA_FP"]
    float_array_fields = ["HOTA", "DetA", "AssA", "DetRe", "DetPr", "AssRe", "AssPr", "LocA", "OWTA"]
    float_fields = ["HOTA(0)", "LocA(0)", "HOTALocA(0)"]

    def __init__(self, compat: bool = True):
        self.compat = compat
        self.fields = self.float_array_fields + self.integer_array_fields + self.float_fields

    # ------------------------------------------------------------------ per sequence
    def eval_sequence(self, data: Dict) -> Dict[str, np.ndarray]:
        nA = len(self.array_labels)
        res = {f: np.zeros(nA) for f in self.float_array_fields + self.integer_array_fields}
        n_trk = int(sum(len(t) for t in data["tracker_ids"]))
        n_gt = int(sum(len(g) for g in data["gt_ids"]))
        if n_trk == 0 or n_gt == 0:                                              # hota.py:36-46
            if n_trk == 0:
                res["HOTA_FN"] = n_gt * np.ones(nA)
            else:
                res["HOTA_FP"] = n_trk * np.ones(nA)
            res["LocA"] = np.ones(nA)
            res.update({"HOTA(0)": 0, "LocA(0)": 1.0, "HOTALocA(0)": 0})
            return res
        sims = [np.asarray(s, dtype=np.float64) for s in data["similarity_scores"]]
        # working copies: the compat path edits id arrays in place, like the reference does to its caller's data
        gts = [np.array(g, dtype=np.int64).reshape(-1, 1) for g in data["gt_ids"]]
        trk = [np.array(t, dtype=np.int64).reshape(-1, 1) for t in data["tracker_ids"]]
        ng, nt = int(data["num_gt_ids"]), int(data["num_tracker_ids"])
        if self.compat:
            pot, g_cnt, t_cnt = self._associate_compat(gts, trk, sims, ng, nt)
        else:
            pot, g_cnt, t_cnt = self._associate(gts, trk, sims, ng, nt)
        align = pot / (g_cnt + t_cnt - pot)                                       # hota.py:100
        matched = np.zeros((nA, ng, nt))
        for g2, tr, sim in zip(gts, trk, sims):
            n, K = len(g2), len(tr)
            if n == 0:                                                           # hota.py:106-109
                res["HOTA_FP"] += K
                continue
            if K == 0:                                                           # hota.py:110-113
                res["HOTA_FN"] += n
                continue
            if self.compat:
                # the frame's alignment block is read from the first K COLUMNS, not the tracker ids; np.squeeze drops
                # EVERY unit axis, so K == 1 or n == 1 frames broadcast differently (hota.py:116-131, kept statement by statement)
                score = np.squeeze(align[g2[:, None], 0:K]) * sim
                try:
                    rows, cols = linear_sum_assignment(-score)
                except ValueError:
                    score = np.squeeze(score)
                    try:
                        rows, cols = linear_sum_assignment(-score)
                    except ValueError:
                        rows, cols = linear_sum_assignment(-score[0, :, :])
            else:
                rows, cols = linear_sum_assignment(-(align[g2[:, 0][:, None], tr[:, 0][None, :]] * sim))   # hota.py:123
            for a, alpha in enumerate(self.array_labels):
                try:                                                             # statement order of hota.py:134-149
                    ok = sim[rows, cols] >= alpha - _EPS
                    r_, c_ = rows[ok], cols[ok]
                    m = len(r_)
                    res["HOTA_TP"][a] += m
                    res["HOTA_FN"][a] += n - m
                    res["HOTA_FP"][a] += K - m
                    if m > 0:
                        res["LocA"][a] += sum(sim[r_, c_])
                        if self.compat:
                            matched[a][g2[r_], tr[c_]] += 1
                        else:
                            matched[a][g2[r_, 0], tr[c_, 0]] += 1
                except IndexError:
                    if not self.compat:
                        raise
                    res["HOTA_FN"][a] += n
                    res["HOTA_FP"][a] += K
        tp = np.maximum(1, res["HOTA_TP"])                                        # hota.py:146-155
        for a in range(nA):
            mc = matched[a]
            res["AssA"][a] = np.sum(mc * (mc / np.maximum(1, g_cnt + t_cnt - mc))) / tp[a]
            res["AssRe"][a] = np.sum(mc * (mc / np.maximum(1, g_cnt))) / tp[a]
            res["AssPr"][a] = np.sum(mc * (mc / np.maximum(1, t_cnt))) / tp[a]
        res["LocA"] = np.maximum(1e-10, res["LocA"]) / np.maximum(1e-10, res["HOTA_TP"])
        return self._compute_final_fields(res)

    @staticmethod
    def _jaccard(sim):
        den = sim.sum(0)[None, :] + sim.sum(1)[:, None] - sim                    # hota.py:58-62
        out = np.zeros_like(sim)
        ok = den > 0 + _EPS
        out[ok] = sim[ok] / den[ok]
        return out

    def _associate(self, gts, trk, sims, ng, nt):
        """Published definition: potential matches and id counts indexed by identity."""
        pot, g_cnt, t_cnt = np.zeros((ng, nt)), np.zeros((ng, 1)), np.zeros((1, nt))
        for g2, tr, sim in zip(gts, trk, sims):
            if len(g2) and len(tr):
                pot[g2[:, 0][:, None], tr[:, 0][None, :]] += self._jaccard(sim)
            g_cnt[g2[:, 0]] += 1
            t_cnt[0, tr[:, 0]] += 1
        return pot, g_cnt, t_cnt

    def _associate_compat(self, gts, trk, sims, ng, nt):
        """The accumulation as the reference's patched first pass executes it (hota.py:52-97): the numpy
        statements are kept one for one because which branch runs depends on numpy's own broadcasting errors."""
        pot, g_cnt, t_cnt = np.zeros((ng, nt)), np.zeros((ng, 1)), np.zeros((1, nt))
        for g2, tr, sim in zip(gts, trk, sims):
            if len(g2) < 1:                                                      # hota.py:54-55
                continue
            g = g2[:, 0]                                                         # a VIEW: `g -= 1` below edits gts[t]
            jac = self._jaccard(sim)
            try:
                pot[g[:, None], tr[None, :]] += jac                              # broadcasts only for K == 1
            except (ValueError, IndexError):
                pot[:len(g), :len(tr)] += jac                                    # by position (hota.py:69-71)
            if len(g_cnt) <= g.max():
                g -= 1
            try:
                g_cnt[g] += 1
            except IndexError:
                pass
            try:
                if len(t_cnt) <= tr.max():                                       # len() of the [1, nt] row vector is 1
                    tr -= tr.min()
            except ValueError:                                                   # empty frame
                pass
            tr -= 1                                                              # hota.py:88; -1 wraps to the last column
            try:
                t_cnt[tr] += 1
            except IndexError:
                t_cnt = t_cnt[0]
                t_cnt[tr] += 1
        return pot, g_cnt, t_cnt

    # ------------------------------------------------------------------ combination (hota.py:166-176, _base_metric.py:200-208)
    def combine_sequences(self, all_res: Dict[str, Dict]) -> Dict[str, np.ndarray]:
        res = {f: sum(r[f] for r in all_res.values()) for f in self.integer_array_fields}
        w = np.maximum(1.0, res["HOTA_TP"])
        for f in ("AssRe", "AssPr", "AssA"):
            res[f] = sum(r[f] * r["HOTA_TP"] for r in all_res.values()) / w
        loca = sum(r["LocA"] * r["HOTA_TP"] for r in all_res.values())
        res["LocA"] = np.maximum(1e-10, loca) / np.maximum(1e-10, res["HOTA_TP"])
        return self._compute_final_fields(res)

    @staticmethod
    def _compute_final_fields(res):                                              # hota.py:213-226
        tp, fn, fp = res["HOTA_TP"], res["HOTA_FN"], res["HOTA_FP"]
        res["DetRe"] = tp / np.maximum(1, tp + fn)
        res["DetPr"] = tp / np.maximum(1, tp + fp)
        res["DetA"] = tp / np.maximum(1, tp + fn + fp)
        res["HOTA"] = np.sqrt(res["DetA"] * res["AssA"])
        res["OWTA"] = np.sqrt(res["DetRe"] * res["AssA"])
        res["HOTA(0)"], res["LocA(0)"] = res["HOTA"][0], res["LocA"][0]
        res["HOTALocA(0)"] = res["HOTA(0)"] * res["LocA(0)"]
        return res


def similarity_scores(gt_boxes: Sequence[np.ndarray], trk_boxes: Sequence[np.ndarray], device="cuda") -> List[np.ndarray]:
    """Per-frame IoU matrices [n_t, K_t] of pixel x0y0x1y1 boxes, all frames of a sequence in one launch
    (`moy_box_iou`, val.py:298-308 + :517-553).  Frames are padded to the sequence maxima on the host."""
    T = len(gt_boxes)
    n = max([len(g) for g in gt_boxes] + [1])
    K = max([len(t) for t in trk_boxes] + [1])
    a = np.zeros((T, n, 4), np.float32)
    b = np.zeros((T, K, 4), np.float32)
    na, nb = np.zeros(T, np.int32), np.zeros(T, np.int32)
    for t in range(T):
        na[t], nb[t] = len(gt_boxes[t]), len(trk_boxes[t])
        a[t, :na[t]] = np.asarray(gt_boxes[t], np.float32).reshape(-1, 4)
        b[t, :nb[t]] = np.asarray(trk_boxes[t], np.float32).reshape(-1, 4)
    dev = torch.device(device)
    iou = ops.box_iou(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev), torch.from_numpy(na).to(dev),
                      torch.from_numpy(nb).to(dev)).cpu().numpy()
    return [iou[t, :na[t], :nb[t]].astype(np.float64) for t in range(T)]


def build_hota_data(gt_ids: Sequence[np.ndarray], trk_ids: Sequence[np.ndarray], sims: Sequence[np.ndarray]) -> Dict:
    """The dictionary the validator hands to `eval_sequence` (val.py:291-310): ids re-indexed to 0..N-1 and shaped
    [n, 1] (the layout the reference's evaluator needs)."""
    ug = np.unique(np.concatenate([np.asarray(g, np.int64).reshape(-1) for g in gt_ids] + [np.zeros(0, np.int64)]))
    ut = np.unique(np.concatenate([np.asarray(t, np.int64).reshape(-1) for t in trk_ids] + [np.zeros(0, np.int64)]))
    return {
        "num_timesteps": len(sims), "num_gt_ids": len(ug), "num_tracker_ids": max(1, len(ut)),
        "num_gt_dets": int(sum(len(g) for g in gt_ids)), "num_tracker_dets": int(sum(len(t) for t in trk_ids)),
        "gt_ids": [np.searchsorted(ug, np.asarray(g, np.int64).reshape(-1)).reshape(-1, 1) for g in gt_ids],
        "tracker_ids": [np.searchsorted(ut, np.asarray(t, np.int64).reshape(-1)).reshape(-1, 1) for t in trk_ids],
        "similarity_scores": list(sims),
    }


class TrackValidator:
    """Sequence-wise validation loop (val.py:185-507 for the tracking half): run the predictor over the frames of each
    sequence, keep the rows that carry a track id (`obj_idxes >= 0`, val.py:383-386), optionally write MOT-style txt
    (engine/results.py:475-512) and score every sequence with HOTA; `results["COMBINED"]` = `combine_sequences`.

    `sequences`: iterable of dicts {name, frames: uint8 [T, H, W, 3] BGR (or float [T, 3, H, W]),
    gt_boxes: T arrays [n, 4] pixel x0y0x1y1, gt_ids: T int arrays [n]}."""

    def __init__(self, predictor, compat: bool = True, chunk: Optional[int] = None, save_dir: Optional[str] = None):
        self.predictor, self.metric, self.save_dir = predictor, HOTA(compat=compat), save_dir
        self.chunk = chunk or predictor.batch

    def eval_tracks(self, gt_boxes, gt_ids, trk_boxes, trk_ids) -> Dict[str, np.ndarray]:
        sims = similarity_scores(gt_boxes, trk_boxes, device=self.predictor.device)
        return self.metric.eval_sequence(build_hota_data(gt_ids, trk_ids, sims))

    def __call__(self, sequences: Iterable[Dict]) -> Dict[str, Dict[str, np.ndarray]]:
        import os
        out: Dict[str, Dict[str, np.ndarray]] = {}
        for seq in sequences:
            frames, T = seq["frames"], len(seq["gt_boxes"])
            trk_boxes, trk_ids = [], []
            for s in range(0, T, self.chunk):
                for r in self.predictor(frames[s:s + self.chunk]):
                    if r.track_id is None:                                        # detection-style fallback: no active track
                        trk_boxes.append(np.zeros((0, 4), np.float32)); trk_ids.append(np.zeros(0, np.int64))
                        continue
                    k = min(len(r.track_id), len(r.boxes))                        # track_id is not conf-filtered (predict.py:61-76)
                    trk_boxes.append(r.boxes[:k, :4].astype(np.float32)); trk_ids.append(np.asarray(r.track_id[:k], np.int64))
                    if self.save_dir:
                        os.makedirs(self.save_dir, exist_ok=True)
                        r.save_txt(os.path.join(self.save_dir, f"{seq['name']}.txt"))
            out[seq["name"]] = self.eval_tracks(seq["gt_boxes"], seq["gt_ids"], trk_boxes, trk_ids)
        if out:
            out["COMBINED"] = self.metric.combine_sequences({k: v for k, v in out.items()})
        return out

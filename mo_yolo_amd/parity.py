"""Parity statistics between two runs of the tracking step on the same frames (engine vs engine, engine vs oracle outputs).

Query ROWS of two runs are comparable only through the encoder token they were selected from (`topk_ind`): a lower
precision run may rank near-tied tokens differently, which permutes rows without changing what is detected.  So rows are
matched by token; reported per call:
  topk_overlap                 share of selected tokens common to both runs (mean over frames)
  box_max_err_matched          max |cxcywh difference| over matched rows
  score_max_err_matched        max |score difference| over matched rows
  hs_max_err_matched           max |difference| of the decoder output embedding over matched rows (the well-conditioned
                               quantity: the fixture score head amplifies it ~100x)
  birth_flip_frac_of_active    matched rows whose `obj_idxes >= 0` differs / rows active in the reference run
  ids_equal                    obj_idxes identical as arrays (same order, same ids)
"""
from __future__ import annotations

import torch


def engine_pair_stats(got, want, nq: int):
    tg, tw = got["topk_ind"].long().cpu(), want["topk_ind"].long().cpu()
    B = tg.shape[0]
    bg, bw = got["boxes"].float().cpu(), want["boxes"].float().cpu()
    sg, sw = got["scores"].float().cpu(), want["scores"].float().cpu()
    ig, iw = got["obj_idxes"].cpu(), want["obj_idxes"].cpu()
    overlap, box_err, score_err, flips, active_w, active_g, matched = 0.0, 0.0, 0.0, 0, 0, 0, 0
    hs_err = 0.0
    hg = got["hs"].float().cpu() if "hs" in got and "hs" in want else None
    hw = want["hs"].float().cpu() if hg is not None else None
    order_same = 0
    for b in range(B):
        pos_w = {int(t): i for i, t in enumerate(tw[b].tolist())}
        rows_g, rows_w = [], []
        for i, t in enumerate(tg[b].tolist()):
            j = pos_w.get(int(t))
            if j is not None:
                rows_g.append(i)
                rows_w.append(j)
        overlap += len(rows_g) / nq
        order_same += int(torch.equal(tg[b], tw[b]))
        active_w += int((iw[b] >= 0).sum())
        active_g += int((ig[b] >= 0).sum())
        if rows_g:
            rg, rw = torch.tensor(rows_g), torch.tensor(rows_w)
            matched += len(rows_g)
            box_err = max(box_err, float((bg[b, rg] - bw[b, rw]).abs().max()))
            score_err = max(score_err, float((sg[b, rg] - sw[b, rw]).abs().max()))
            flips += int(((ig[b, rg] >= 0) != (iw[b, rw] >= 0)).sum())
            if hg is not None:
                hs_err = max(hs_err, float((hg[b, rg] - hw[b, rw]).abs().max()))
    return {
        "frames": B, "topk_overlap": round(overlap / B, 5), "topk_order_equal_frames": order_same, "rows_matched": matched,
        "box_max_err_matched": box_err, "score_max_err_matched": score_err, "hs_max_err_matched": hs_err,
        "births_flipped": flips, "active_rows_reference": active_w, "active_rows": active_g,
        "birth_flip_frac_of_active": round(flips / max(1, active_w), 5),
        "ids_equal": bool(torch.equal(ig, iw)),
    }

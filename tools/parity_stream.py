#!/usr/bin/env python3
"""Parity study over long synthetic streams (600-frame MOT17-shape sequences), run on the GPU box:

  * the fp32 engine, free running, is held to the CPU oracle on every `--oracle-every`-th frame (top-k order, logits, ids);
  * the bf16 and fp16 engines, free running, are compared with the fp32 engine on EVERY frame: top-k overlap, max box / score /
    decoder-output error over rows matched by selected token, births flipped as a fraction of the active rows;
  * HOTA (reference evaluator's algorithm, mo_yolo_amd.evaluate.HOTA(compat=True), and the published definition) of every
    engine's tracks against the synthetic ground truth, and the difference to the fp32 engine.

Writes one JSON document (default profiles/parity_r02.json).  `python tools/parity_stream.py --frames 600 --seqs 0 1`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from mo_yolo_amd import evaluate as E  # noqa: E402
from mo_yolo_amd.engine import TrackEngine  # noqa: E402
from mo_yolo_amd.fixtures import fixture  # noqa: E402
from mo_yolo_amd.parity import engine_pair_stats  # noqa: E402
from mo_yolo_amd.synth import SyntheticSequence, to_network_input  # noqa: E402


def tracks_of(out, b, W, H):
    act = (out["obj_idxes"][b] >= 0)
    bx = out["boxes"][b][act]
    xyxy = torch.stack([(bx[:, 0] - bx[:, 2] / 2) * W, (bx[:, 1] - bx[:, 3] / 2) * H, (bx[:, 0] + bx[:, 2] / 2) * W,
                        (bx[:, 1] + bx[:, 3] / 2) * H], -1)
    return xyxy.numpy().astype(np.float32), out["obj_idxes"][b][act].numpy().astype(np.int64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2")
    ap.add_argument("--frames", type=int, default=600)
    ap.add_argument("--seqs", type=int, nargs="+", default=[0, 1])
    ap.add_argument("--batch", type=int, default=24)
    ap.add_argument("--oracle-every", type=int, default=50)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "parity_r02.json"))
    a = ap.parse_args()
    torch.set_num_threads(min(16, os.cpu_count() or 1))      # (the oracle: eager CPU ops collapse with hundreds of threads)
    cfg, arch, sd = fixture(a.config)
    H, W, B = cfg["H"], cfg["W"], a.batch
    dev = "cuda"
    engines = {"f32": TrackEngine(arch, sd, H, W, batch=B, dtype=torch.float32), "bf16": TrackEngine(arch, sd, H, W, batch=B, dtype=torch.bfloat16),
               "f16": TrackEngine(arch, sd, H, W, batch=B, dtype=torch.float16)}
    doc = {"config": a.config, "frames_per_sequence": a.frames, "sequences": a.seqs, "engine_batch": B,
           "note": "rows are matched by selected encoder token (mo_yolo_amd/parity.py); every engine runs free (its own top-k)"}
    agg = {k: dict(frames=0, overlap=0.0, order_equal=0, box=0.0, score=0.0, hs=0.0, flips=0, active_ref=0, active=0, ids_equal_frames=0,
                   masked=0) for k in ("bf16", "f16")}
    oracle = dict(frames=0, topk_equal=0, logits_max_err=0.0, ids_exact=0, topk_overlap_min=1.0, box_max_err_matched=0.0,
                  score_max_err_matched=0.0, hs_max_err_matched=0.0, births_flipped=0, active_rows_oracle=0)
    hota = {}
    t_start = time.time()
    for sid in a.seqs:
        seq = SyntheticSequence(sid, H, W, cfg["style"])
        gt_boxes, gt_ids = zip(*[seq.boxes(t) for t in range(a.frames)])
        trk = {k: ([], []) for k in engines}
        for t0 in range(0, a.frames, B):
            n = min(B, a.frames - t0)
            fr = seq.frames(t0, n)
            if n < B:
                fr = np.concatenate([fr, np.repeat(fr[-1:], B - n, 0)])
            x = torch.from_numpy(fr).to(dev)
            outs = {}
            for k, e in engines.items():
                o = e.forward(x)
                torch.cuda.synchronize()
                outs[k] = {kk: v[:n].cpu().clone() for kk, v in o.items() if hasattr(v, "shape") and v.shape[:1] == (B,)}
            for b in range(n):
                for k in engines:
                    bx, ids = tracks_of(outs[k], b, W, H)
                    trk[k][0].append(bx); trk[k][1].append(ids)
            for k in ("bf16", "f16"):
                st = engine_pair_stats(outs[k], outs["f32"], arch.nq)
                g = agg[k]
                g["frames"] += n; g["overlap"] += st["topk_overlap"] * n; g["order_equal"] += st["topk_order_equal_frames"]
                g["box"] = max(g["box"], st["box_max_err_matched"]); g["score"] = max(g["score"], st["score_max_err_matched"])
                g["hs"] = max(g["hs"], st["hs_max_err_matched"]); g["flips"] += st["births_flipped"]
                g["active_ref"] += st["active_rows_reference"]; g["active"] += st["active_rows"]
                g["masked"] += int(outs[k]["n_masked"].sum())
            # the CPU oracle as the checker of the fp32 engine on a sample of the stream
            for b in range(n):
                t = t0 + b
                if t % a.oracle_every:
                    continue
                from oracle import track_oracle as O
                with torch.no_grad():
                    r = O.forward(to_network_input(fr[b:b + 1]), sd, arch)
                same = bool(torch.equal(outs["f32"]["topk_ind"][b].long(), r["topk_ind"][0]))
                sc_o = r["dec_scores"].sigmoid().max(-1).values
                st = engine_pair_stats({kk: v[b:b + 1] for kk, v in outs["f32"].items()},
                                       dict(topk_ind=r["topk_ind"], boxes=r["dec_bboxes"], scores=sc_o, obj_idxes=O.assign_ids(sc_o), hs=r["hs"]), arch.nq)
                oracle["topk_overlap_min"] = min(oracle["topk_overlap_min"], st["topk_overlap"])
                for kk in ("box_max_err_matched", "score_max_err_matched", "hs_max_err_matched"):
                    oracle[kk] = max(oracle[kk], st[kk])
                oracle["births_flipped"] += st["births_flipped"]; oracle["active_rows_oracle"] += st["active_rows_reference"]
                oracle["frames"] += 1
                oracle["topk_equal"] += int(same)
                if same:
                    oracle["logits_max_err"] = max(oracle["logits_max_err"], float((outs["f32"]["logits"][b] - r["dec_scores"][0]).abs().max()))
                    oracle["ids_exact"] += int(torch.equal(outs["f32"]["obj_idxes"][b], O.assign_ids(r["dec_scores"][0].sigmoid().max(-1).values)))
            print(f"[parity] seq {sid} frames {t0 + n}/{a.frames}  ({time.time() - t_start:.0f} s)", flush=True)
        for k in engines:
            sims = E.similarity_scores(gt_boxes, trk[k][0], device=dev)
            data = E.build_hota_data(gt_ids, trk[k][1], sims)
            for name, metric in (("compat", E.HOTA(compat=True)), ("published", E.HOTA(compat=False))):
                res = metric.eval_sequence({kk: (list(v) if isinstance(v, list) else v) for kk, v in data.items()})
                hota.setdefault(k, {}).setdefault(name, {})[f"seq{sid}"] = {m: float(np.mean(res[m])) for m in ("HOTA", "DetA", "AssA")}
            hota[k].setdefault("tracks_per_frame", {})[f"seq{sid}"] = float(np.mean([len(i) for i in trk[k][1]]))
    for k, g in agg.items():
        n = max(1, g["frames"])
        doc[k + "_vs_f32_engine"] = {
            "frames": g["frames"], "topk_overlap_mean": round(g["overlap"] / n, 5), "topk_order_equal_frames": g["order_equal"],
            "box_max_err_matched": g["box"], "score_max_err_matched": g["score"], "hs_max_err_matched": g["hs"],
            "births_flipped": g["flips"], "active_rows_f32": g["active_ref"], "active_rows": g["active"],
            "birth_flip_frac_of_active": round(g["flips"] / max(1, g["active_ref"]), 5), "masked_tokens_selected": g["masked"]}
    oracle["note"] = ("frames of the free-running stream, NOT the margin fixtures: adjacent encoder scores of unconstrained frames come as close "
                      "as 1e-6 relative, so two correct fp32 evaluations rank a few near-ties differently (topk_equal < frames); the rows "
                      "themselves agree (matched by token)")
    doc["f32_engine_vs_cpu_oracle"] = oracle
    doc["hota"] = hota
    d = {}
    for k in ("bf16", "f16"):
        for name in ("compat", "published"):
            d[f"{k}.{name}"] = {s: round(100 * (hota[k][name][s]["HOTA"] - hota["f32"][name][s]["HOTA"]), 4) for s in hota["f32"][name]}
    doc["hota_delta_points_vs_f32"] = d
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps({k: doc[k] for k in ("bf16_vs_f32_engine", "f16_vs_f32_engine", "f32_engine_vs_cpu_oracle", "hota_delta_points_vs_f32")}))


if __name__ == "__main__":
    main()

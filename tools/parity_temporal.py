#!/usr/bin/env python3
"""Agreement study of the TEMPORAL mode (carried track queries) over long synthetic streams -- the only
setting in which ids persist across frames, so the only one in which "HOTA within 0.1 of the reference" is a falsifiable
sentence (VERDICT r3 #3a; reference anchors: ultralytics/utils/hota.py:24-164, nn/modules/head.py:206-221, 1232-1237).

  * `--seqs` sequences run in lockstep (batch element = sequence) for `--frames` frames on an fp32, a bf16 and an fp16
    temporal engine, every engine free running from a reset;
  * per sequence: HOTA / DetA / AssA of the 16-bit engine's tracks scored AGAINST THE fp32 ENGINE'S TRACKS as ground truth
    (100 = identical; HOTA matches ids by association, so a renumbering alone costs nothing -- a flipped birth, a lost or a
    swapped track does);
  * the fp32 engine against `oracle/temporal_oracle.py` (CPU, spec-parity: the reference's carried branch cannot run,
    SURVEY section 0.3) on the first `--oracle-seqs` sequences: ids exact frame by frame while they are, agreement-HOTA over
    the whole stream;
  * slot occupancy: live tracks per frame, `n_overflow` (active rows beyond the slots: dropped, counted, never silent).

    python tools/parity_temporal.py --slots 200 --frames 200 --seqs 0 1 2 3 --out profiles/parity_r04_temporal_c2.json
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

from mo_yolo_amd.engine import TrackEngine  # noqa: E402
from mo_yolo_amd.fixtures import fixture  # noqa: E402
from mo_yolo_amd.parity import _xyxy, agreement_hota  # noqa: E402
from mo_yolo_amd.synth import SyntheticSequence, to_network_input  # noqa: E402


def tracks(boxes, ids, W, H):
    act = ids >= 0
    return _xyxy(boxes[act], W, H).numpy().astype("float32"), ids[act].numpy().astype("int64")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2")
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--seqs", type=int, nargs="+", default=[0, 1, 2, 3])
    ap.add_argument("--slots", type=int, default=200)
    ap.add_argument("--nq", type=int, default=None, help="detect queries per frame (default: the fixture's; weights do not depend on it)")
    ap.add_argument("--oracle-seqs", type=int, default=1)
    ap.add_argument("--oracle-frames", type=int, default=None)
    ap.add_argument("--birth", type=float, default=0.4, help="score_thresh (head.py:1146 ships 0.4)")
    ap.add_argument("--miss", type=float, default=0.5, help="filter_score_thresh (ships 0.5)")
    ap.add_argument("--tolerance", type=int, default=5, help="miss_tolerance (ships 5)")
    ap.add_argument("--dtypes", nargs="+", default=["bf16", "f16"])
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "parity_r04_temporal_c2.json"))
    a = ap.parse_args()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    cfg, arch, sd = fixture(a.config)
    if a.nq:
        import dataclasses
        arch = dataclasses.replace(arch, nq=a.nq)
    H, W, B, T, nm, nq = cfg["H"], cfg["W"], len(a.seqs), a.frames, a.slots, arch.nq
    dev = "cuda"
    dts = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}
    kw = dict(temporal=nm, score_thresh=a.birth, filter_score_thresh=a.miss, miss_tolerance=a.tolerance)
    engines = {k: TrackEngine(arch, sd, H, W, batch=B, dtype=dts[k], **kw) for k in ["f32"] + a.dtypes}
    seqs = [SyntheticSequence(s, H, W, cfg["style"]) for s in a.seqs]
    trk = {k: [[] for _ in range(B)] for k in engines}
    live = {k: np.zeros((T, B), np.int32) for k in engines}
    over = {k: np.zeros((T, B), np.int32) for k in engines}
    births = {k: np.zeros((T, B), np.int32) for k in engines}
    f32_rows = [[] for _ in range(B)]                  # per frame: (ids over [tracks | detect], n_in) for the oracle comparison
    t_start = time.time()
    for t in range(T):
        x = torch.from_numpy(np.concatenate([s.frames(t, 1) for s in seqs])).to(dev)
        for k, e in engines.items():
            n_before = e.trk["n"].clone()
            o = e.forward(x)
            torch.cuda.synchronize()
            ids, bx = o["obj_idxes"].cpu(), o["boxes"].float().cpu()
            live[k][t] = o["n_tracks"].cpu().numpy()
            over[k][t] = o["n_overflow"].cpu().numpy()
            births[k][t] = (ids[:, nm:] >= 0).sum(1).numpy()
            for b in range(B):
                trk[k][b].append(tracks(bx[b], ids[b], W, H))
                if k == "f32":
                    f32_rows[b].append((ids[b].clone(), int(n_before[b]), bx[b].clone(), o["scores"][b].cpu().clone()))
        if (t + 1) % 20 == 0:
            print(f"[temporal parity] frame {t + 1}/{T}  live f32 {live['f32'][t].tolist()}  overflow so far "
                  f"{over['f32'][:t + 1].sum(0).tolist()}  ({time.time() - t_start:.0f} s)", flush=True)
    doc = {"config": a.config, "frames_per_sequence": T, "sequences": a.seqs, "slots": nm, "queries": nq,
           "live_tracks_per_frame_f32": live["f32"].tolist(),
           "thresholds": {"birth": a.birth, "miss": a.miss, "miss_tolerance": a.tolerance},
           "note": "temporal mode (carried track queries, DESIGN.md section 7); every engine free running from a reset; agreement = the "
                   "engine's tracks scored against the fp32 temporal engine's tracks as ground truth (100 = identical)"}
    occ = {}
    for k in engines:
        occ[k] = {"live_tracks_mean": round(float(live[k].mean()), 2), "live_tracks_max": int(live[k].max()),
                  "live_tracks_last_frame": live[k][-1].tolist(), "births_per_frame_mean": round(float(births[k].mean()), 2),
                  "n_overflow_total": int(over[k].sum()), "frames_with_overflow": int((over[k] > 0).sum()),
                  "frames_saturated": int((live[k] >= nm).sum())}
    doc["slot_occupancy"] = occ
    agree = {}
    for k in a.dtypes:
        per = {f"seq{s}": agreement_hota(trk[k][b], trk["f32"][b], device=dev) for b, s in enumerate(a.seqs)}
        m = {name: {f: round(float(np.mean([per[q][name][f] for q in per])), 3) for f in ("HOTA", "DetA", "AssA")}
             for name in ("compat", "published")}
        agree[k] = {"per_sequence": per, "mean": m,
                    "min": {name: {f: round(float(np.min([per[q][name][f] for q in per])), 3) for f in ("HOTA", "DetA", "AssA")}
                            for name in ("compat", "published")}}
        # simple id-free figures beside HOTA: live-count difference and first frame at which the id sets differ
        first_diff = []
        for b in range(B):
            fd = next((t for t in range(T) if not np.array_equal(np.sort(trk[k][b][t][1]), np.sort(trk["f32"][b][t][1]))), None)
            first_diff.append(fd)
        agree[k]["first_frame_with_a_different_id_set"] = first_diff
        agree[k]["live_tracks_abs_diff_mean"] = round(float(np.abs(live[k].astype(np.int64) - live["f32"]).mean()), 3)
    doc["agreement_vs_f32_temporal_engine"] = agree

    # ---- fp32 engine vs the CPU oracle of the spec
    if a.oracle_seqs:
        from oracle.temporal_oracle import TemporalOracle
        To = a.oracle_frames or T
        res = {}
        for b, s in list(enumerate(a.seqs))[:a.oracle_seqs]:
            orc = TemporalOracle(sd, arch, nm)
            orc_trk, ids_exact_frames, first_id_diff, max_box, max_score = [], 0, None, 0.0, 0.0
            t0 = time.time()
            for t in range(To):
                w = orc.step(to_network_input(seqs[b].frames(t, 1)), orig_hw=(H, W))
                orc_trk.append(tracks(w["boxes"], w["ids"], W, H))
                ids_e, n_in, bx_e, sc_e = f32_rows[b][t]
                idx = list(range(n_in)) + list(range(nm, nm + nq))
                same = (n_in == w["n_in"]) and ids_e[idx].tolist() == w["ids"].tolist()
                if same:
                    ids_exact_frames += 1
                    act = w["ids"] >= 0                         # (rows without an id may be permuted among themselves by top-k near-ties)
                    if bool(act.any()):
                        ii = torch.tensor(idx)[act]
                        max_box = max(max_box, float((bx_e[ii] - w["boxes"][act]).abs().max()))
                        max_score = max(max_score, float((sc_e[ii] - w["scores"][act]).abs().max()))
                elif first_id_diff is None:
                    first_id_diff = t
                if (t + 1) % 20 == 0:
                    print(f"[temporal parity] oracle seq {s} frame {t + 1}/{To} ({time.time() - t0:.0f} s)", flush=True)
            res[f"seq{s}"] = {"frames": To, "frames_ids_exact_in_row_order": ids_exact_frames, "first_frame_ids_differ": first_id_diff,
                              "box_max_err_active_rows_while_exact": max_box, "score_max_err_active_rows_while_exact": max_score,
                              "agreement_hota_engine_vs_oracle_tracks": agreement_hota(trk["f32"][b][:To], orc_trk, device=dev)}
        doc["f32_temporal_engine_vs_cpu_oracle"] = res
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps({"slot_occupancy": occ, "agreement_mean": {k: v["mean"] for k, v in agree.items()},
                      "oracle": doc.get("f32_temporal_engine_vs_cpu_oracle")}))


if __name__ == "__main__":
    main()

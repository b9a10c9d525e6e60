#!/usr/bin/env python3
"""Probe: where does a 16-bit engine's error enter, and is it one-sided?

Runs the fp32 engine and a 16-bit engine of the same weights on the same frames, the 16-bit engine with the fp32 engine's
query selection injected (so every row is the same token and seams compare row by row), stepping both plans launch by launch
and snapshotting every seam: backbone / neck layer outputs, feats, the value planes, per decoder layer attn / e1 / offsets+weights /
samp / layer output / refined boxes, hs and the final logits.  Per seam it reports

    rms        rms of the fp32 values
    err_rms    rms of (16-bit - fp32)
    err_mean   SIGNED mean of (16-bit - fp32)            one-sided errors show up here
    bias_ratio |err_mean| / err_rms                      ~ 1/sqrt(n) for unbiased rounding noise
    common     rms over channels of the row-mean error   (the part of the error every query shares)

and for the logits the split  logit_err = w . (common hs error) + w . (per-row hs error).

    python tools/probes/seam_bias.py --config c4 --frames 2 --dtype bf16 --out profiles/r03_seam_bias_c4.json
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from mo_yolo_amd.engine import TrackEngine  # noqa: E402
from mo_yolo_amd.fixtures import fixture  # noqa: E402
from mo_yolo_amd.synth import SyntheticSequence  # noqa: E402


def walk(eng, frames, topk=None):
    """Run the plan launch by launch; returns {seam: fp32 CPU tensor}."""
    snap = {}
    meta = eng.meta
    names = [m["name"] for m in meta]
    i_topk = eng._topk_step
    i_mha = [i for i, n in enumerate(names) if n.startswith("mha_core")]
    i_msda = [i for i, n in enumerate(names) if n.startswith("msda_fused")]
    nl = len(i_msda)
    eng.input.copy_(frames)
    grab = lambda t: t.detach().float().cpu().clone()
    # which buffers the decoder loop reuses: find them through the plan's public handles
    pos = 0

    def run_to(stop):
        nonlocal pos
        if stop > pos:
            eng.run_steps(pos, stop)
            torch.cuda.synchronize()
            pos = stop
    run_to(i_topk)
    for li, v in sorted(eng.layer_views.items()):
        if v is not None and li not in eng.virtual_layers:
            snap[f"bb.L{li:02d}"] = grab(v.tensor())
    snap["feats"] = grab(eng.feats.tensor())
    B, S = eng.B, eng.S
    vp = eng.value_planes
    snap["value.layer0"] = grab(vp[:eng.arch.nh * B * S])
    snap["value.layer5"] = grab(vp[(nl - 1) * eng.arch.nh * B * S:nl * eng.arch.nh * B * S])
    snap["enc_scores_all"] = grab(eng.scores_all)
    run_to(i_topk + 1)
    if topk is not None:
        tl = topk.to(eng.dev, torch.int32).reshape(B, -1)
        eng.topk_local.copy_(tl)
        eng.topk_global.copy_(tl + torch.arange(B, device=eng.dev, dtype=torch.int32)[:, None] * S)
    snap["topk"] = eng.topk_local.cpu().clone()
    run_to(i_mha[0] - 1)                                   # up to (not including) the first q|k|v GEMM
    snap["embed0"] = grab(eng.features.tensor())
    snap["refer_logit"] = grab(eng.refer_logit)
    snap["query_pos"] = grab(eng.query_pos.tensor())
    for l in range(nl):
        run_to(i_msda[l] + 1)
        dv = eng.debug_views                               # handles of the buffers the decoder loop reuses
        for k in ("attn", "e1", "offaw", "samp"):
            snap[f"dec{l}.{k}"] = grab(dv[k])
        run_to(i_mha[l + 1] - 1 if l + 1 < nl else len(names))   # (nothing after the last layer overwrites its outputs)
        emb, ref = eng.layer_out[l]
        snap[f"dec{l}.out"] = grab(emb.tensor())
        snap[f"dec{l}.ref"] = grab(ref)
    run_to(len(names))
    snap["hs"] = grab(eng.hs.tensor())
    snap["logits"] = grab(eng.logits)
    snap["boxes"] = grab(eng.boxes)
    snap["obj_idxes"] = eng.obj_idxes.cpu().clone()
    return snap


def stats(a, b):
    """a = 16-bit, b = fp32 reference."""
    d = (a - b).double()
    bb = b.double()
    n = d.numel()
    err_rms = float(d.pow(2).mean().sqrt())
    out = dict(n=n, rms=float(bb.pow(2).mean().sqrt()), err_rms=err_rms, err_mean=float(d.mean()),
               bias_ratio=float(abs(d.mean()) / max(err_rms, 1e-30)), err_max=float(d.abs().max()))
    if d.dim() == 2 and d.shape[0] > 1:
        cm = d.mean(0)                                     # the error every row shares, per channel
        out["common_rms"] = float(cm.pow(2).mean().sqrt())
        out["per_row_rms"] = float((d - cm).pow(2).mean().sqrt())
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c4")
    ap.add_argument("--frames", type=int, default=2)
    ap.add_argument("--seq", type=int, default=0)
    ap.add_argument("--t0", type=int, default=0)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16"])
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    cfg, arch, sd = fixture(a.config)
    H, W, B = cfg["H"], cfg["W"], a.frames
    fr = torch.from_numpy(SyntheticSequence(a.seq, H, W, cfg["style"]).frames(a.t0, B)).to("cuda")
    dt = {"bf16": torch.bfloat16, "f16": torch.float16}[a.dtype]
    e32 = TrackEngine(arch, sd, H, W, batch=B, dtype=torch.float32)
    s32 = walk(e32, fr)
    del e32
    e16 = TrackEngine(arch, sd, H, W, batch=B, dtype=dt)
    free = walk(e16, fr)                                   # its own selection: how far does the free-running engine drift
    s16 = walk(e16, fr, topk=s32["topk"])
    doc = {"config": a.config, "dtype": a.dtype, "frames": B, "seq": a.seq, "t0": a.t0,
           "note": "16-bit engine with the fp32 engine's query selection injected; err = 16-bit - fp32", "seams": {}}
    for k in s32:
        if k in ("topk", "obj_idxes") or k not in s16:       # (a fused launch leaves no seam: layer 0 of the 16-bit engines)
            continue
        x, y = s16[k], s32[k]
        if x.shape != y.shape:
            continue
        doc["seams"][k] = stats(x.reshape(-1, x.shape[-1]) if x.dim() > 1 else x, y.reshape(-1, y.shape[-1]) if y.dim() > 1 else y)
    # logit error split: w . common-mode hs error (every row moves together) + w . per-row error
    d = f"model.{len(arch.layers)}.decoder"
    w = sd[f"{d}.dec_score_head.{arch.ndl - 1}.weight"][0].double()
    dh = (s16["hs"] - s32["hs"]).double().view(B, -1, 256)
    cm = dh.mean(1, keepdim=True)
    lg16, lg32 = s16["logits"].double().view(B, -1), s32["logits"].double().view(B, -1)
    doc["logit_split"] = {
        "w_norm": float(w.norm()),
        "logit_err_mean": float((lg16 - lg32).mean()), "logit_err_std": float((lg16 - lg32).std()),
        "from_common_hs_error_per_frame": [float(v) for v in (cm @ w).view(-1)],
        "from_per_row_hs_error_std": float(((dh - cm) @ w).std()),
        "hs_query_deviation_along_w_std": float(((s32["hs"].double().view(B, -1, 256) - s32["hs"].double().view(B, -1, 256).mean(1, keepdim=True)) @ w).std()),
        "active_rows_fp32": int((s32["obj_idxes"] >= 0).sum()), "active_rows_16bit_forced_topk": int((s16["obj_idxes"] >= 0).sum()),
        "active_rows_16bit_free": int((free["obj_idxes"] >= 0).sum()),
        "topk_overlap_free": float(sum(len(set(free["topk"][b].tolist()) & set(s32["topk"][b].tolist())) for b in range(B)) / (B * arch.nq)),
    }
    for k, v in doc["seams"].items():
        extra = f"  common {v['common_rms']:.2e} per-row {v['per_row_rms']:.2e}" if "common_rms" in v else ""
        print(f"{k:16s} rms {v['rms']:.3e} err_rms {v['err_rms']:.2e} err_mean {v['err_mean']:+.2e} bias_ratio {v['bias_ratio']:.3f}{extra}")
    print(json.dumps(doc["logit_split"]))
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(doc, f, indent=1)


if __name__ == "__main__":
    main()

"""GPU parity of the whole per-frame step (TrackEngine over libmoyolo.so) against the CPU oracle and
the committed reference goldens, seam by seam.  fp32 engine: decoder logits within 1e-3 (the
north-star bar), ids/rows exact given the same query order.  bf16 engine: stated looser bars."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from mo_yolo_amd.engine import TrackEngine, PlanOptions
from oracle import track_oracle as O
from tests._util import fixture, frames_u8, golden, net_input

DEV = "cuda"

# Absolute bars (round 4, VERDICT r3 #3c): 1.5 x what THIS test measures on its fixture frames (deterministic: same frames, same
# kernels, no atomics; the values are printed on every run):   (box, decoder output, score -- max abs error over rows matched by
# token --, births flipped on matched rows, top-k overlap)
MEASURED_16 = {
    ("tiny", torch.bfloat16): (1.47e-3, 0.183, 0.0505, 1, 0.9867), ("c2", torch.bfloat16): (2.64e-3, 0.298, 0.1158, 10, 0.9758),
    ("c2", torch.float16): (9.6e-4, 0.0741, 0.0205, 0, 0.9967), ("c4", torch.bfloat16): (2.75e-3, 0.350, 0.0449, 1, 0.982),
    ("c4", torch.float16): (4.9e-4, 0.091, 0.0105, 0, 0.998), ("full", torch.bfloat16): (2.82e-3, 0.260, 0.0539, 1, 0.9667),
    ("full", torch.float16): (1.7e-4, 0.0213, 0.0043, 0, 1.0),
}


def nhwc_to_nchw(view, B, hw):
    h, w = hw
    return view.tensor().float().cpu().view(B, h, w, -1).permute(0, 3, 1, 2)


def run_oracle(cfg, arch, sd, t0, n, topk=None):
    with torch.no_grad():
        x = net_input(cfg, t0, n)
        feats_in, outs = O.backbone_neck(x, sd, arch, return_all=True)
        trace = {}
        r = O.head_forward(feats_in, sd, arch, topk_ind=topk, trace=trace)
    return outs, r, trace


@pytest.mark.parametrize("name,B", [("tiny", 2), ("tiny3", 3)])
def test_engine_fp32_seams_vs_oracle(name, B):
    cfg, arch, sd = fixture(name)
    eng = TrackEngine(arch, sd, cfg["H"], cfg["W"], batch=B, dtype=torch.float32, input_format="u8")
    fr = torch.from_numpy(frames_u8(cfg, 0, B)).to(DEV)
    out = eng.forward(fr)
    torch.cuda.synchronize()
    outs, r, trace = run_oracle(cfg, arch, sd, 0, B)
    errs = []

    def chk(label, got, want, atol):
        e = float((got - want).abs().max())
        if not (e <= atol):
            errs.append(f"{label}: max err {e:.3e} > {atol}")

    assert eng.virtual_layers == {10, 11, 13, 14}          # the neck's Upsample + Concat pairs are folded into their C2f
    for L_ in arch.layers:
        if L_.i in eng.virtual_layers:
            continue
        chk(f"L{L_.i}({L_.kind})", nhwc_to_nchw(eng.layer_views[L_.i], B, eng.layer_hw[L_.i]), outs[L_.i], 1e-4)
    S = eng.S
    chk("feats", eng.feats.tensor().float().cpu().view(B, S, -1), r["feats"], 1e-4)
    chk("enc_scores_all", eng.scores_all.cpu().view(B, S, -1), r["enc_scores_all"], 2e-4)
    assert np.array_equal(eng.valid.cpu().numpy().astype(bool), r["valid"][0, :, 0].numpy())
    tk = out["topk_ind"].cpu().long()
    # enc_output is materialised for the selected tokens only (the pass over all S tokens yields the score logits)
    chk("features[selected]", eng.features.tensor().float().cpu().view(B, arch.nq, -1),
        r["features"][torch.arange(B)[:, None], tk], 2e-4)
    if not torch.equal(tk, r["topk_ind"]):
        errs.append(f"topk order differs at {(tk != r['topk_ind']).sum().item()} positions")
        outs, r, trace = run_oracle(cfg, arch, sd, 0, B, topk=tk)      # same query order for the rest
    assert out["n_masked"].cpu().tolist() == [0] * B
    chk("refer_bbox_logit", out["refer_bbox_logit"].cpu(), r["refer_bbox_logit"], 2e-4)
    chk("query_pos", eng.query_pos.tensor().float().cpu().view(B, arch.nq, -1), r["query_pos"], 5e-4)
    for li in range(arch.ndl):
        e, ref = eng.layer_out[li]
        # engine buffers ping-pong: only the last two layers are still live after the step
        if li >= arch.ndl - 2:
            chk(f"dec{li}.out", e.tensor().float().cpu().view(B, arch.nq, -1), trace[li]["out"], 5e-4)
            chk(f"dec{li}.refined", ref.cpu().view(B, arch.nq, 4), trace[li]["refined"], 1e-4)
    chk("logits", out["logits"].cpu(), r["dec_scores"], 1e-3)                   # north-star bar
    chk("boxes", out["boxes"].cpu(), r["dec_bboxes"], 1e-4)
    chk("y", out["y"].cpu(), r["y"], 1e-4)
    assert not errs, "\n".join(errs)
    scores = r["dec_scores"].sigmoid().max(-1).values
    for b in range(B):
        ids = O.assign_ids(scores[b])
        assert torch.equal(out["obj_idxes"][b].cpu(), ids), f"frame {b} ids"
        rows, tid = O.postprocess(r["y"][b], r["dec_scores"][b], ids, 0.25, orig_hw=(cfg["H"], cfg["W"]))
        n = int(out["n_rows"][b])
        assert n == rows.shape[0]
        assert torch.allclose(out["rows"][b, :n].cpu(), rows, atol=2e-2)        # pixels
        if tid is not None:
            assert torch.equal(out["track_id"][b, :int(out["n_ids"][b])].cpu(), tid)


@pytest.mark.parametrize("name", ["tiny", "tiny3"])
def test_engine_fp32_vs_reference_goldens_stream(name):
    """Directly against what the REFERENCE produced (tests/golden): y, scores, per-frame ids."""
    cfg, arch, sd = fixture(name)
    g = golden(name)
    T = cfg["frames"]
    eng = TrackEngine(arch, sd, cfg["H"], cfg["W"], batch=T, dtype=torch.float32)
    out = eng.forward(torch.from_numpy(frames_u8(cfg, 0, T)).to(DEV))
    torch.cuda.synchronize()
    assert np.array_equal(out["topk_ind"][0].cpu().numpy(), g["t0.topk_ind"].reshape(-1))
    assert np.allclose(out["y"].cpu().numpy(), g["y"], atol=2e-4)
    assert np.allclose(out["scores"].cpu().numpy(), g["scores"], atol=2e-4)
    assert float(g["score_margin"]) > 1e-3
    assert np.array_equal(out["obj_idxes"].cpu().numpy(), g["obj_idxes"])       # bit-exact ids
    for t in range(T):
        n = int(out["n_rows"][t])
        assert np.allclose(out["rows"][t, :n].cpu().numpy(), g[f"post.{t}.boxes"], atol=5e-2, rtol=1e-5)
        if bool(g[f"post.{t}.is_track"]):
            k = int(out["n_ids"][t])
            assert np.array_equal(out["track_id"][t, :k].cpu().numpy(), g[f"post.{t}.track_id"].reshape(-1))


def _direct_vs_golden(name, B, **ekw):
    """Free-running fp32 engine (its own top-k) against the reference's outputs on EVERY fixture frame, directly: same query
    selection in the same order, y within 1e-3 (logit scale too), obj_idxes == the reference's obj_idxes.  The fixtures carry
    the margins of SURVEY App. G (adjacent top-k scores > 1e-3 apart, boundary > 5e-3, no score within 1e-2 of 0.4 / 0.5)."""
    cfg, arch, sd = fixture(name)
    g = golden(name)
    T = cfg["frames"]
    assert float(g["topk_min_gap_all"]) > 1e-3 and float(g["topk_boundary_gap_all"]) > 5e-3 and float(g["score_margin"]) > 1e-2
    assert int(g["n_masked_in_topk"]) == 0
    eng = TrackEngine(arch, sd, cfg["H"], cfg["W"], batch=B, dtype=torch.float32, **ekw)
    logit = lambda p: np.log(np.clip(p, 1e-7, 1 - 1e-7) / (1 - np.clip(p, 1e-7, 1 - 1e-7)))
    worst = 0.0
    for t0 in range(0, T, B):
        out = eng.forward(torch.from_numpy(frames_u8(cfg, t0, B)).to(DEV))
        torch.cuda.synchronize()
        assert out["n_masked"].cpu().tolist() == [0] * B
        for b in range(min(B, T - t0)):
            t = t0 + b
            assert np.array_equal(out["topk_ind"][b].cpu().numpy(), g["topk_ind_all"][t]), f"frame {t}: query selection / order"
            y = out["y"][b].cpu().numpy()
            worst = max(worst, float(np.abs(y - g["y"][t]).max()))
            assert np.allclose(y, g["y"][t], atol=1e-3), (t, np.abs(y - g["y"][t]).max())
            assert np.allclose(logit(y[:, 4:]), logit(g["y"][t][:, 4:]), atol=1e-3), f"frame {t}: decoder logits"
            assert np.array_equal(out["obj_idxes"][b].cpu().numpy(), g["obj_idxes"][t]), f"frame {t}: track ids"
            n = int(out["n_rows"][b])
            assert np.allclose(out["rows"][b, :n].cpu().numpy(), g[f"post.{t}.boxes"], atol=5e-2, rtol=1e-5), f"frame {t}: predictor rows"
            if bool(g[f"post.{t}.is_track"]):
                k = int(out["n_ids"][b])
                assert np.array_equal(out["track_id"][b, :k].cpu().numpy(), g[f"post.{t}.track_id"].reshape(-1))
    print(f"[{name}{' split_f16' if ekw else ''}] {T} frames direct vs reference: max |y diff| {worst:.2e}")


def test_engine_fp32_full_scale_yaml_vs_reference_golden():
    """`yolo_track.yaml` at its OWN scale (depth 1.0 / width 1.0: 46 M parameters, 3 / 6 / 6 / 3 bottlenecks, 256- and 512-channel
    3x3 convolutions, head inputs 256 / 512 / 512 -- the model the reference's entry script builds, start_train.py:11) at a small
    resolution, free running, directly against the reference's outputs: none of the shape-specialised kernels (conv_ws, c2f_fused,
    the N = 256 weight-stationary forms) applies at these widths, so this pins the general paths."""
    _direct_vs_golden("full", 2)
    cfg, arch, sd = fixture("full")
    eng = TrackEngine(arch, sd, cfg["H"], cfg["W"], batch=2, dtype=torch.bfloat16)
    names = [m["name"] for m in eng.meta]
    assert not any(n.startswith(("c2f fused", "stem+conv1")) for n in names), names      # (those forms are s-scale shapes)


def test_engine_fp32_c2_vs_reference_golden():
    """Config C2 (s-scale, 1088x608, nq 300): all 8 golden frames, direct comparison."""
    _direct_vs_golden("c2", 4)


@pytest.mark.parametrize("name,B", [("tiny", 3), ("c2", 4), ("c4", 2), ("full", 2)])
def test_engine_split_f16_products_vs_reference_goldens(name, B):
    """Round 5: the fp32 engine with every `moy_gemm` product in split fp16 precision (`split_f16=True`, MOY_F32X3) held
    to the SAME bar as the exact-fp32 engine, directly against the reference's outputs on every fixture frame: same query selection in
    the same order, y and decoder logits within 1e-3, obj_idxes and track ids bit-exact, predictor rows -- north_star's parity sentence,
    at several times the exact engine's matrix rate."""
    _direct_vs_golden(name, B, split_f16=True)


@pytest.mark.slow
def test_engine_fp32_c4_vs_reference_golden():
    """Config C4 (1920x1088, nq 500): both golden frames, direct comparison."""
    _direct_vs_golden("c4", 2)


def test_engine_bench_scale_kernel_paths_vs_small_batch():
    """at B <= 4 (B*S < 65536) moy_gemm never takes the weight-stationary kernels, the head-plane value layout or
    the output row remap of input_proj, and the 3x3 convs stay on the per-tile kernels.  B = 6 at the C2 shape crosses those
    thresholds (B*S = 81 396 rows; 6 x 45 conv tiles): the fp32 engine's query selection is the reference (previous test), and
    the 16-bit engines at B = 6 must agree with the SAME dtype at B = 2 row for row -- every kernel is batch-invariant by
    construction except for the kernel choice, so any difference is a bench-scale-only path going wrong."""
    cfg, arch, sd = fixture("c2")
    B = 6
    fr = torch.from_numpy(frames_u8(cfg, 0, B)).to(DEV)
    # (scores: the fixture's last score head amplifies the decoder output ~100x, DESIGN.md section 2 -- hs is the tight check)
    from mo_yolo_amd.parity import engine_pair_stats
    f32 = TrackEngine(arch, sd, cfg["H"], cfg["W"], batch=2, dtype=torch.float32)
    ref = []
    for t0 in range(0, B, 2):
        ref.append({k: v.clone() for k, v in f32.forward(fr[t0:t0 + 2]).items()})
    del f32
    for dt in (torch.bfloat16, torch.float16):
        big = TrackEngine(arch, sd, cfg["H"], cfg["W"], batch=B, dtype=dt)
        ob = {k: v.clone() for k, v in big.forward(fr).items()}
        small = TrackEngine(arch, sd, cfg["H"], cfg["W"], batch=2, dtype=dt)
        for i, t0 in enumerate(range(0, B, 2)):
            os_ = small.forward(fr[t0:t0 + 2])
            torch.cuda.synchronize()
            sub = {k: v[t0:t0 + 2] for k, v in ob.items() if hasattr(v, "shape") and v.shape[:1] == (B,)}
            st = engine_pair_stats(sub, os_, arch.nq)                 # bench-scale kernels vs small-batch kernels, same dtype
            sf = engine_pair_stats(sub, ref[i], arch.nq)              # ... vs fp32: what the dtype itself costs on these frames
            # (a random-init network amplifies a one-ulp difference in a summation order like any other rounding, DESIGN.md
            # section 2: two correct 16-bit kernel paths differ by about as much as either differs from fp32 -- a bench-scale-only
            # path going WRONG shows up as a multiple of that)
            assert st["topk_overlap"] > 0.9, st
            for k in ("box_max_err_matched", "hs_max_err_matched", "score_max_err_matched"):
                assert st[k] <= 3.0 * sf[k] + 1e-6, (dt, k, st, sf)    # (two noise realisations: sqrt(2) in rms, more in the max)
            mb, mh = MEASURED_16[("c2", dt)][:2]                      # (six fixture frames here, four there: 2 x instead of 1.5 x)
            assert sf["box_max_err_matched"] < 2.0 * mb and sf["hs_max_err_matched"] < 2.0 * mh, (dt, sf)


@pytest.mark.parametrize("B", [104, 16])
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_engine_folded_input_proj_plan_vs_classic_plan(dt, B, monkeypatch):
    """Round 4: at bench scale (every pyramid level >= 65536 rows per launch: B >= 102 at the C2 shape; round 6: from 8192 rows, B >= 13,
    `PlanOptions.fold_min_rows` -- 16 frames is the second case) input_proj (Conv1x1 + BN, no
    activation, head.py:838-839) is folded into its two linear consumers -- value projection and score pass read P3/P4/P5 directly
    with composed weights, the projected features exist for the nq selected tokens only.  Same function, one rounding fewer (the
    bf16 feature map is never formed), so the folded engine must sit at least as close to fp32 as the classic plan does, and the two
    16-bit plans must agree with each other as two noise realisations do (previous test's argument)."""
    from mo_yolo_amd.parity import engine_pair_stats
    cfg, arch, sd = fixture("c2")
    fr = torch.from_numpy(frames_u8(cfg, 0, B)).to(DEV)
    classic = TrackEngine(arch, sd, cfg["H"], cfg["W"], batch=B, dtype=dt, options=PlanOptions(fold_proj=False))
    folded = TrackEngine(arch, sd, cfg["H"], cfg["W"], batch=B, dtype=dt)
    assert folded.fold_proj and not classic.fold_proj and folded.feats is None
    assert sum("valid-runs" in m["name"] for m in folded.meta) == 3 and not any("level_" in m["name"] for m in classic.meta)
    oc = {k: v.clone() for k, v in classic.forward(fr).items()}
    of = {k: v.clone() for k, v in folded.forward(fr).items()}
    torch.cuda.synchronize()
    S, nq = folded.S, arch.nq
    # the value maps and the score logits of ALL tokens: composed 16-bit weights against (16-bit weights o 16-bit features)
    # (round 5: the folded plan samples level 0 raw -- `value_tokens` -- so its planes hold the tokens of levels 1.. only)
    Sv = folded.value_tokens
    assert (Sv == S - 76 * 136) == (folded.p3raw is not None) and classic.value_tokens == S
    vc = classic.value_planes.view(-1, B, S, 32)[:, :, S - Sv:].float()
    vf = folded.value_planes.view(-1, B, Sv, 32).float()
    sc, sf_ = classic.scores_all, folded.scores_all
    eps = 2.0 ** -8 if dt == torch.bfloat16 else 2.0 ** -11
    dv = (vc - vf).abs()
    if float(dv.max()) > 16 * eps * float(vc.abs().max()):
        # (round 4 saw 3.26 on 12.2 here once: the first-tile wait count of the DMA rings, DESIGN.md section 4 round 5 item 1 -- the
        #  report of WHERE stays, so that anything of the kind names its kernel)
        bad = (dv > 16 * eps * float(vc.abs().max())).nonzero()
        where = dict(outliers=len(bad), planes=sorted(set(bad[:, 0].tolist()))[:16], frames=sorted(set(bad[:, 1].tolist()))[:16],
                     tokens=(int(bad[:, 2].min()) + S - Sv, int(bad[:, 2].max()) + S - Sv), level_starts=(0, 76 * 136, 76 * 136 + 38 * 68),
                     backbone_layers_that_differ=[i for i, v in classic.layer_views.items() if v is not None and i not in classic.virtual_layers
                                                  and not torch.equal(v.tensor(), folded.layer_views[i].tensor())])
        raise AssertionError(f"value planes of the two plans differ by {float(dv.max()):.4f} (max |v| {float(vc.abs().max()):.3f}): {where}")
    assert float((sc - sf_).abs().max()) <= 16 * eps * max(1.0, float(sc.abs().max())), float((sc - sf_).abs().max())
    # the selected tokens' projected features: fp32 product rounded once, against the classic plan's feature map rows
    tk = of["topk_ind"].long()
    rows = (torch.arange(B, device=DEV)[:, None] * S + tk).flatten()
    fc = classic.feats.tensor().float()[rows]
    ff = folded.feats_selected.tensor().float()
    assert float((fc - ff).abs().max()) <= 2 * eps * max(1.0, float(fc.abs().max()))
    f32 = TrackEngine(arch, sd, cfg["H"], cfg["W"], batch=2, dtype=torch.float32)
    worst = {"c": {}, "f": {}, "cf": {}}
    for t0 in range(0, B, 2):
        r = f32.forward(fr[t0:t0 + 2])
        torch.cuda.synchronize()
        sub = lambda o: {k: v[t0:t0 + 2] for k, v in o.items() if hasattr(v, "shape") and v.shape[:1] == (B,)}
        for key, st in (("c", engine_pair_stats(sub(oc), r, nq)), ("f", engine_pair_stats(sub(of), r, nq)),
                        ("cf", engine_pair_stats(sub(of), sub(oc), nq))):
            for k in ("box_max_err_matched", "hs_max_err_matched", "score_max_err_matched"):
                worst[key][k] = max(worst[key].get(k, 0.0), st[k])
            worst[key]["topk_overlap"] = min(worst[key].get("topk_overlap", 1.0), st["topk_overlap"])
    print(f"[fold] {dt}: classic vs fp32 {worst['c']}  folded vs fp32 {worst['f']}  folded vs classic {worst['cf']}")
    for k in ("box_max_err_matched", "hs_max_err_matched", "score_max_err_matched"):
        assert worst["f"][k] <= 1.5 * worst["c"][k] + 1e-6, (k, worst)           # (max over 104 frames of two noise realisations)
        assert worst["cf"][k] <= 3.0 * worst["c"][k] + 1e-6, (k, worst)
    assert worst["f"]["topk_overlap"] >= worst["c"]["topk_overlap"] - 0.02 and worst["cf"]["topk_overlap"] > 0.9, worst


@pytest.mark.parametrize("name,B,dt", [("c2", 104, torch.bfloat16), ("c2", 104, torch.float16), ("c4", 34, torch.bfloat16)])
def test_engine_level0_sampled_raw_vs_projected_planes(name, B, dt, monkeypatch):
    """Round 5: level 0 of the deformable attention gathered raw and projected after the bilinear sum (`moy_msda_raw0`; the P3 value
    planes are never formed) against the same folded plan WITH those planes (`PlanOptions(p3_raw=False)`).  Same function, the level-0 contribution
    rounded once (the gathered vector) instead of once per projected value: the two plans must agree like two noise realisations of
    the type, and the raw plan must sit as close to the fp32 engine as the other does; everything in front of the decoder is the
    same launches -- bit-identical scores of all tokens, same query selection, same planes of the other levels."""
    from mo_yolo_amd.parity import engine_pair_stats
    cfg, arch, sd = fixture(name)
    fr = torch.from_numpy(frames_u8(cfg, 0, B)).to(DEV)
    proj = TrackEngine(arch, sd, cfg["H"], cfg["W"], batch=B, dtype=dt, options=PlanOptions(p3_raw=False))
    raw = TrackEngine(arch, sd, cfg["H"], cfg["W"], batch=B, dtype=dt)
    assert proj.fold_proj and raw.fold_proj and proj.p3raw is None and raw.p3raw is not None
    hw0 = raw.shapes[0][0] * raw.shapes[0][1]
    # (one value launch fewer, one `moy_query_order` more: round 6, the gather walks a frame's queries in Morton order)
    assert raw.value_tokens == raw.S - hw0 and raw.num_launches == proj.num_launches - 1 + int(raw.qperm is not None)
    assert sum(m["name"].startswith("msda_raw0") for m in raw.meta) == arch.ndl and not any(m["name"].startswith("msda_raw0") for m in proj.meta)
    op = {k: v.clone() for k, v in proj.forward(fr).items()}
    orw = {k: v.clone() for k, v in raw.forward(fr).items()}
    torch.cuda.synchronize()
    S, Sv = raw.S, raw.value_tokens
    assert torch.equal(proj.scores_all, raw.scores_all) and torch.equal(op["topk_ind"], orw["topk_ind"])
    assert torch.equal(proj.value_planes.view(-1, B, S, 32)[:, :, hw0:], raw.value_planes.view(-1, B, Sv, 32))
    f32 = TrackEngine(arch, sd, cfg["H"], cfg["W"], batch=2, dtype=torch.float32)
    worst = {"p": {}, "r": {}, "rp": {}}
    for t0 in range(0, min(B, 24), 2):
        r = f32.forward(fr[t0:t0 + 2])
        torch.cuda.synchronize()
        sub = lambda o: {k: v[t0:t0 + 2] for k, v in o.items() if hasattr(v, "shape") and v.shape[:1] == (B,)}
        for key, st in (("p", engine_pair_stats(sub(op), r, arch.nq)), ("r", engine_pair_stats(sub(orw), r, arch.nq)),
                        ("rp", engine_pair_stats(sub(orw), sub(op), arch.nq))):
            for k in ("box_max_err_matched", "hs_max_err_matched", "score_max_err_matched"):
                worst[key][k] = max(worst[key].get(k, 0.0), st[k])
            worst[key]["births_flipped"] = worst[key].get("births_flipped", 0) + st["births_flipped"]
    print(f"[p3raw] {name} {dt}: planes vs fp32 {worst['p']}  raw vs fp32 {worst['r']}  raw vs planes {worst['rp']}")
    for k in ("box_max_err_matched", "hs_max_err_matched", "score_max_err_matched"):
        assert worst["r"][k] <= 1.5 * worst["p"][k] + 1e-6, (k, worst)           # (max over the frames of two noise realisations)
        assert worst["rp"][k] <= 3.0 * worst["p"][k] + 1e-6, (k, worst)
    assert worst["r"]["births_flipped"] <= 1.5 * worst["p"]["births_flipped"] + 3, worst


def test_engine_forked_value_projection_bit_identical_eager_and_graph(monkeypatch):
    """Round 4: the P3 value projection on a side stream / half of the compute units beside the P4 / P5 branch of the neck
    (engine.py `_plan_fork`; moy_set_cu_limit only changes how many persistent blocks walk the same row tiles): outputs, value
    planes and scores bit-identical to the plan on one stream, eagerly and replayed from the hipGraph (two parallel branches)."""
    cfg, arch, sd = fixture("c2")
    B = 104
    fr = torch.from_numpy(frames_u8(cfg, 0, B)).to(DEV)
    # (the fork is a property of the plan that still projects level 0)
    plain = TrackEngine(arch, sd, cfg["H"], cfg["W"], batch=B, dtype=torch.bfloat16, options=PlanOptions(p3_raw=False))
    forked = TrackEngine(arch, sd, cfg["H"], cfg["W"], batch=B, dtype=torch.bfloat16, options=PlanOptions(p3_raw=False, fork_value=128))
    assert plain._fork is None and forked._fork is not None and forked.fold_proj
    fk = forked._fork
    assert forked.meta[fk["side"]]["name"].startswith("gemm1x1") and "N1536" in forked.meta[fk["side"]]["name"]
    assert fk["after"] < fk["side"] < fk["join"] and forked.meta[fk["join"]]["name"].startswith("msda")
    keys = ("scores", "boxes", "obj_idxes", "topk_ind", "hs", "rows", "n_rows")
    op = {k: v.clone() for k, v in plain.forward(fr).items()}
    vp, sp = plain.value_planes.clone(), plain.scores_all.clone()
    of = {k: v.clone() for k, v in forked.forward(fr).items()}
    torch.cuda.synchronize()
    assert torch.equal(vp, forked.value_planes) and torch.equal(sp, forked.scores_all)
    for k in keys:
        assert torch.equal(op[k], of[k]), k
    forked.value_planes.zero_()
    forked.capture()
    og = forked.forward(fr)
    torch.cuda.synchronize()
    assert torch.equal(vp, forked.value_planes)
    for k in keys:
        assert torch.equal(op[k], og[k]), ("graph", k)
    assert forked.lib.moy_set_cu_limit(0) == 0          # the budget is back to the whole device after a pass


def test_engine_fp16_c5_batched_sequences_graph():
    """Config C5: fp16 activations/weights (the reference's own `half` switch, predictor.py:131), frames of
    4 sequences batched into one hipGraph-captured step; boxes/scores vs the oracle per sequence."""
    cfg, arch, sd = fixture("tiny")
    nseq, per = 4, 2
    eng = TrackEngine(arch, sd, cfg["H"], cfg["W"], batch=nseq * per, dtype=torch.float16)
    fr = np.concatenate([frames_u8(cfg, 0, per, seq_id=s) for s in range(nseq)])
    frd = torch.from_numpy(fr).to(DEV)
    eng.forward(frd)
    eng.capture()
    out = eng.forward(frd)
    torch.cuda.synchronize()
    with torch.no_grad():
        x = torch.cat([net_input(cfg, 0, per, seq_id=s) for s in range(nseq)])
        r = O.forward(x, sd, arch)
    # rows matched by selected token (fp16 reorders near-tied encoder scores)
    tk, rk = out["topk_ind"].cpu().long(), r["topk_ind"]
    ref_scores = r["dec_scores"].sigmoid().max(-1).values
    db, ds, matched = 0.0, 0.0, 0
    for b_ in range(nseq * per):
        pos = {int(t): i for i, t in enumerate(rk[b_])}
        for i, t in enumerate(tk[b_]):
            j = pos.get(int(t))
            if j is not None:
                matched += 1
                db = max(db, float((out["boxes"][b_, i].cpu() - r["dec_bboxes"][b_, j]).abs().max()))
                ds = max(ds, float((out["scores"][b_, i].cpu() - ref_scores[b_, j]).abs()))
    frac = matched / tk.numel()
    print(f"[fp16 C5] token overlap {frac:.3f} box err {db:.4f} score err {ds:.4f}")
    assert frac > 0.9
    assert float(db) < 0.02 and float(ds) < 0.1


def _eager_16bit_oracle(cfg, arch, sd, x_u8, dt):
    """The oracle executed in `dt` by eager torch on the GPU: model and input cast to 16 bits, every op eager -- the reference's
    own `half` switch (engine/predictor.py:131, nn/autobackend.py:108).  The YARDSTICK of the 16-bit engines: what the
    arithmetic type costs on this network without any of this repository's kernels."""
    sdh = {k: (v.to(DEV, dt) if v.is_floating_point() else v.to(DEV)) for k, v in sd.items()}
    from mo_yolo_amd.synth import to_network_input
    with torch.no_grad():
        r = O.forward(to_network_input(x_u8).to(dt), sdh, arch, anchor_dtype=torch.float32)
    sc = r["dec_scores"].float().sigmoid().max(-1).values.cpu()
    return dict(topk_ind=r["topk_ind"].cpu(), boxes=r["dec_bboxes"].float().cpu(), scores=sc, obj_idxes=O.assign_ids(sc), hs=r["hs"].float().cpu())


@pytest.mark.parametrize("name,dt", [("tiny", torch.bfloat16), ("c2", torch.bfloat16), ("c2", torch.float16), ("c4", torch.bfloat16),
                                     ("c4", torch.float16), ("full", torch.bfloat16), ("full", torch.float16)])
def test_engine_16bit_within_the_budget_of_the_arithmetic_type(name, dt):
    """16-bit activations / weights with fp32 accumulation, FREE RUNNING (own top-k), on the fixture frames against the pinned fp32
    oracle -- and next to it the oracle itself run in the same 16-bit type by eager torch (`_eager_16bit_oracle`).  A random-init
    network amplifies rounding noise (0.3 % per stored activation grows to 2.5 % at the feature maps whatever
    executes it), so 'close to fp32' is a statement about the TYPE; what is asserted about the KERNELS is that the engine is never
    further from fp32 than eager torch in that type, on boxes, decoder output, scores, flipped births and selected tokens.
    Rows are matched by selected encoder token (mo_yolo_amd/parity.py)."""
    from mo_yolo_amd.parity import engine_pair_stats, token_id_agreement
    cfg, arch, sd = fixture(name)
    B = min(4, cfg["frames"])
    eng = TrackEngine(arch, sd, cfg["H"], cfg["W"], batch=B, dtype=dt)
    fr = torch.from_numpy(frames_u8(cfg, 0, B)).to(DEV)
    with torch.no_grad():
        r = O.forward(net_input(cfg, 0, B), sd, arch)
    out = eng.forward(fr)
    torch.cuda.synchronize()
    assert torch.isfinite(out["y"]).all() and int(out["n_masked"].sum()) == 0
    sc = r["dec_scores"].sigmoid().max(-1).values
    want = dict(topk_ind=r["topk_ind"], boxes=r["dec_bboxes"], scores=sc, obj_idxes=O.assign_ids(sc), hs=r["hs"])
    got = {k: v.clone() for k, v in out.items() if hasattr(v, "shape") and v.shape[:1] == (B,)}
    yard = _eager_16bit_oracle(cfg, arch, sd, fr, dt)
    st, sy = engine_pair_stats(got, want, arch.nq), engine_pair_stats(yard, want, arch.nq)
    tk, ty = token_id_agreement(got, want, arch.nq), token_id_agreement(yard, want, arch.nq)
    print(f"[{dt} {name}] engine {st} {tk}\n[{dt} {name}] eager torch {sy} {ty}")
    for k in ("box_max_err_matched", "hs_max_err_matched", "score_max_err_matched"):
        assert st[k] <= sy[k] * 1.0 + 1e-6, (k, st[k], sy[k])
    assert st["births_flipped"] <= sy["births_flipped"] + 1, (st, sy)
    assert st["topk_overlap"] >= sy["topk_overlap"] - max(0.005, 1.5 / arch.nq), (st, sy)     # (1.5 tokens: nq = 60 at "full")
    assert tk["tokens_id_equal_frac"] >= ty["tokens_id_equal_frac"] - 0.02, (tk, ty)
    mb, mh, ms, mf, mo = MEASURED_16[(name, dt)]
    assert st["box_max_err_matched"] <= 1.5 * mb and st["hs_max_err_matched"] <= 1.5 * mh and st["score_max_err_matched"] <= 1.5 * ms, st
    assert st["births_flipped"] <= max(1, math.ceil(1.5 * mf)), st
    assert 1.0 - st["topk_overlap"] <= 1.5 * (1.0 - mo) + 1.0 / arch.nq, st
    if dt == torch.float16 and name in ("c2", "c4"):
        # the fixtures keep every score 0.03 away from the birth / miss thresholds (0.125 in the logit; fp16 moves a logit by 0.014):
        # no birth may flip on a matched row.  The id NUMBERS follow the encoder-score order of the active tokens, which no
        # conditioning of the score head makes 16-bit-proof (measured: tests/golden/make_golden.py:separate_topk) -- reported.
        assert st["births_flipped"] == 0, st


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_decoder_plan_variants_are_bit_identical(dt, monkeypatch):
    """Round-3 plan changes that must not change a bit of the outputs.  (i) q | k and v as two plain products over the
    `x + query_pos` the previous layer's tail wrote (`PlanOptions.qkv_split`; at bench scale they take the weight-stationary kernel) instead
    of one product with a second A operand, on the tiny fixture (forced: 2).  (ii) the score pass over the valid tokens only
    (`PlanOptions.score_runs`) at the C2 shape with enough rows for the weight-stationary score kernel (B x S >= 65 536)."""
    def run(name, B, opt):
        cfg, arch, sd = fixture(name)
        fr = torch.from_numpy(frames_u8(cfg, 0, B)).to(DEV)
        eng = TrackEngine(arch, sd, cfg["H"], cfg["W"], batch=B, dtype=dt, options=PlanOptions(**opt))
        out = {k: v.clone() for k, v in eng.forward(fr).items()}
        torch.cuda.synchronize()
        return out, [m["name"] for m in eng.meta], arch
    keys = ("topk_ind", "logits", "boxes", "hs", "obj_idxes", "y", "rows")
    a, na, arch = run("tiny", 3, dict(qkv_split=0, qkv_fuse_small=False))
    b, nb, _ = run("tiny", 3, dict(qkv_split=2, qkv_fuse_small=False))
    assert len(nb) == len(na) + (arch.ndl - 1), (len(na), len(nb))                  # one more launch per layer after the first
    for k in keys:
        assert torch.equal(a[k], b[k]), ("qkv split", k)
    # (iii, round 5) the next layer's q | k | v projected by the fused tail itself (`PlanOptions.qkv_fuse`): one launch fewer per layer after the first
    f, nf, _ = run("tiny", 3, dict(qkv_split=2, qkv_fuse=True, qkv_fuse_small=False))
    assert len(nf) == len(na) - (arch.ndl - 1) and sum("decoder_tail+qkv" in n for n in nf) == arch.ndl - 1, nf
    for k in keys:
        assert torch.equal(a[k], f[k]), ("qkv fused into the tail", k)
    c, nc_, _ = run("c2", 5, dict(qkv_split=0, score_runs=False))
    d, nd, _ = run("c2", 5, dict(qkv_split=0, score_runs=True))
    assert not any("valid-runs" in n for n in nc_) and sum("valid-runs 7317/13566" in n for n in nd) == 1, nd
    for k in keys:
        assert torch.equal(c[k], d[k]), ("score runs", k)


@pytest.mark.parametrize("name,B,dt,temporal", [("tiny", 3, torch.bfloat16, 0), ("c2", 4, torch.float16, 0), ("c2", 4, torch.bfloat16, 40)])
def test_small_batch_plan_forked_value_and_fused_qkv_bit_identical(name, B, dt, temporal):
    """Round 6, the small-batch leg (one frame of each of a few live sequences per hipGraph replay): the value projection of the classic
    plan on a side stream beside the query selection / first self-attention (`PlanOptions.fork_small_value`: measured +33 us at four frames,
    so off by default -- the chip-filling value launch only takes turns with the chain it was meant to hide behind), the next layer's
    q | k | v projected by the decoder tail (`qkv_fuse_small`: one launch less per layer) and the tail's 32-row blocks change no bit of the
    outputs -- eagerly and replayed from the hipGraph (the fork becomes two parallel branches) -- against the plan without the two."""
    cfg, arch, sd = fixture(name)
    fr = torch.from_numpy(frames_u8(cfg, 0, B)).to(DEV)
    kw = dict(temporal=temporal) if temporal else {}
    plain = TrackEngine(arch, sd, cfg["H"], cfg["W"], batch=B, dtype=dt, options=PlanOptions(fork_small_value=0, qkv_fuse_small=False), **kw)
    small = TrackEngine(arch, sd, cfg["H"], cfg["W"], batch=B, dtype=dt, options=PlanOptions(fork_small_value=262144), **kw)
    assert plain._fork is None and small._fork is not None and small._fork["side_cus"] == 0
    assert small.meta[small._fork["side"]]["name"].startswith("gemm1x1") and small.meta[small._fork["join"]]["name"].startswith("msda")
    assert small.num_launches == plain.num_launches - (arch.ndl - 1) and sum("decoder_tail+qkv" in m["name"] for m in small.meta) == arch.ndl - 1
    # (temporal mode: `rows` beyond n_rows keep whatever an earlier step left there -- the capture's warm-up steps in the replayed engine)
    keys = ("scores", "boxes", "obj_idxes", "topk_ind", "hs", "n_rows", "logits", "y") + (() if temporal else ("rows",))
    op = {k: v.clone() for k, v in plain.forward(fr).items()}
    torch.cuda.synchronize()
    os_ = {k: v.clone() for k, v in small.forward(fr).items()}
    torch.cuda.synchronize()
    for k in keys:
        assert torch.equal(op[k], os_[k]), ("eager", k)
    assert torch.equal(plain.value_planes, small.value_planes)
    small.capture()
    if temporal:
        small.reset_sequence()
    for rep in range(3):
        og = {k: v.clone() for k, v in small.forward(fr).items()}
        torch.cuda.synchronize()
        if temporal and rep:
            break                       # (carried state: only the first replay from a reset equals the eager first step)
        for k in keys:
            assert torch.equal(op[k], og[k]), ("graph replay", rep, k)


def test_engine_graph_replay_matches_eager():
    cfg, arch, sd = fixture("tiny")
    B = 2
    eng = TrackEngine(arch, sd, cfg["H"], cfg["W"], batch=B, dtype=torch.float32)
    fr = torch.from_numpy(frames_u8(cfg, 0, B)).to(DEV)
    a = {k: v.clone() for k, v in eng.forward(fr).items()}
    eng.capture()
    eng.input.zero_()
    b = eng.forward(fr)
    torch.cuda.synchronize()
    for k in ("y", "obj_idxes", "rows", "n_rows", "topk_ind"):
        assert torch.equal(a[k], b[k]), k


@pytest.mark.parametrize("mode", ["per_frame", "temporal", "side_state", "detect"])
def test_every_launch_has_byte_accounting(mode):
    """bench.py's plan_bytes sums `meta["bytes"]` over the launches: no launch kind may be left at zero."""
    if mode == "detect":
        from mo_yolo_amd.config import build_detect_arch
        from mo_yolo_amd.weights import make_fixture_state_dict
        arch = build_detect_arch()
        sd, cfg, kw = make_fixture_state_dict(arch, 5), dict(H=640, W=640), {}
    else:
        cfg, arch, sd = fixture("tiny")
        kw = dict(temporal=8) if mode == "temporal" else dict(side_state=True) if mode == "side_state" else {}
    eng = TrackEngine(arch, sd, cfg["H"], cfg["W"], batch=2, dtype=torch.bfloat16, **kw)
    assert len(eng.meta) == eng.num_launches
    missing = [m["name"] for m in eng.meta if not m["bytes"] > 0]
    assert not missing, missing


def test_streamed_engines_equal_single_engine():
    """Sub-batches on separate HIP streams + graphs (engine.StreamedEngines) give the frames' own results."""
    from mo_yolo_amd.engine import StreamedEngines
    cfg, arch, sd = fixture("tiny")
    fr = torch.from_numpy(frames_u8(cfg, 0, 3)).to(DEV)
    fr = torch.cat([fr, fr[:1]], 0)                                   # 4 frames -> 2 streams x 2
    ref = TrackEngine(arch, sd, cfg["H"], cfg["W"], batch=4, dtype=torch.float32)
    want = {k: v.clone() for k, v in ref.forward(fr).items()}
    pipe = StreamedEngines(arch, sd, cfg["H"], cfg["W"], batch=4, streams=2, graph=True, dtype=torch.float32)
    for _ in range(3):                                               # first call captures, later calls replay
        pipe.forward(fr)
    pipe.synchronize()
    got = pipe.outputs()
    assert torch.equal(got["obj_idxes"], want["obj_idxes"])
    assert torch.equal(got["topk_ind"], want["topk_ind"])
    assert torch.allclose(got["y"], want["y"], atol=1e-6)
    assert torch.equal(got["n_rows"], want["n_rows"])


def test_agreement_hota_against_the_oracle_tracks():
    """BASELINE metric, HOTA half, as a figure that CAN FAIL: HOTA / DetA / AssA of the build's tracks scored AGAINST THE ORACLE'S
    TRACKS as ground truth (100 = identical tracks; evaluator = the path's own, ultralytics/utils/hota.py:24-164), on the 8 fixture
    frames, every engine free running.  (HOTA against the synthetic scene is ~0 for every engine -- a random-init decoder does not
    localise -- so 'within 0.1 of the reference' was true by construction in round 2.)
    fp32: 100 / 100 / 100 exactly.  16-bit: the detection half (DetA) is held to bars = the stream measurements of
    profiles/parity_r03_c2.json minus a margin; the association half is REPORTED only: the reference hands ids out as the rank of a
    row among the active rows of its frame (head.py:1232-1237), so one flipped birth renumbers every later row of that frame."""
    from mo_yolo_amd.parity import agreement_hota, tracks_of
    cfg, arch, sd = fixture("c2")
    T = 8
    fr = torch.from_numpy(frames_u8(cfg, 0, T)).to(DEV)
    with torch.no_grad():
        r = O.forward(net_input(cfg, 0, T), sd, arch)
    ref_out = dict(obj_idxes=O.assign_ids(r["dec_scores"].sigmoid().max(-1).values), boxes=r["dec_bboxes"])
    ref_trk = [tracks_of(ref_out, t, cfg["W"], cfg["H"]) for t in range(T)]
    assert min(len(t[1]) for t in ref_trk) > 0
    res = {}
    for dt in (torch.float32, torch.float16, torch.bfloat16):
        eng = TrackEngine(arch, sd, cfg["H"], cfg["W"], batch=T, dtype=dt)
        out = {k: v.clone() for k, v in eng.forward(fr).items()}
        torch.cuda.synchronize()
        res[dt] = agreement_hota([tracks_of(out, t, cfg["W"], cfg["H"]) for t in range(T)], ref_trk, device=DEV)
        print(f"[agreement-HOTA vs oracle tracks, {dt}] {res[dt]}")
    for kind in ("compat", "published"):
        assert res[torch.float32][kind] == {"HOTA": 100.0, "DetA": 100.0, "AssA": 100.0}, res[torch.float32]
        # measured (deterministic): fp16 HOTA 85.3 / DetA 99.1 / AssA 73.5, bf16 45.8 / 79.4 / 26.5.  Bars = 100 - 1.5 x (100 - measured)
        # where that is tighter than round 3's (DetA bf16: kept at 75); the association half of the per-frame mode is a RANK
        # comparison (one flipped birth renumbers every later row of its frame), so its bf16 bar is a floor, not a quality claim
        assert res[torch.float16][kind]["DetA"] >= 98.6 and res[torch.float16][kind]["AssA"] >= 60.0 and res[torch.float16][kind]["HOTA"] >= 78.0, res[torch.float16]
        assert res[torch.bfloat16][kind]["DetA"] >= 75.0 and res[torch.bfloat16][kind]["AssA"] >= 15.0, res[torch.bfloat16]


def test_side_state_copy_filter_and_fsqm_vs_oracle():
    """a16 (copy half) + a17: the filtered/renumbered copy and the FSQM memory, kept on device, equal the
    oracle's restatement of head.py:1245-1283 / fsqm.py over a multi-frame stream (state carries)."""
    cfg, arch, sd = fixture("tiny")
    sd = dict(sd)
    d = f"model.{len(arch.layers)}.decoder"
    sd[d + f".dec_score_head.{arch.ndl - 1}.bias"] = sd[d + f".dec_score_head.{arch.ndl - 1}.bias"] + 6.0   # many births, some > 0.7
    B, steps = 3, 3
    eng = TrackEngine(arch, sd, cfg["H"], cfg["W"], batch=B, dtype=torch.float32, side_state=True)
    fs = O.FSQMOracle(300, 256)
    for s in range(steps):
        out = eng.forward(torch.from_numpy(frames_u8(cfg, s * B, B)).to(DEV))
        torch.cuda.synchronize()
        sc, bx, ids, hs = (out[k].cpu() for k in ("scores", "boxes", "obj_idxes", "hs"))
        for b in range(B):
            rows, nid = O.tracker_update_copy(sc[b].tolist(), bx[b].numpy(), ids[b].tolist())
            n = int(eng.n_copy[b])
            assert n == len(rows), (s, b, n, len(rows))
            assert eng.copy_rows[b, :n].cpu().tolist() == rows
            assert eng.copy_ids[b, :n].cpu().tolist() == nid
            if rows:
                det = (sc[b][rows].numpy(), bx[b][rows].numpy(), hs[b][rows].float().numpy())
            else:       # no active row: update() hands FSQM the full Instances (head.py:1249-1251)
                det = (sc[b].numpy(), bx[b].numpy(), hs[b].float().numpy())
            fs.online_update(det[0], det[1], det[2], sc[b].numpy(), bx[b].numpy(), ids[b].numpy())
        f = {k: v.cpu().numpy() for k, v in eng.fsqm.items()}
        assert np.array_equal(f["ids"], fs.ids), s
        assert np.array_equal(f["low"], fs.low)
        assert np.allclose(f["conf"], fs.conf) and np.allclose(f["boxes"], fs.boxes)
        assert np.allclose(f["mem"], fs.mem, atol=1e-6)
        head, cnt, ovf = f["pool_hc"]
        assert ovf == 0 and cnt == len(fs.pool)
        assert [int(f["pool"][(head + k) % len(f["pool"])]) for k in range(cnt)] == fs.pool
    assert (fs.ids >= 0).sum() > 0, "fixture should inject at least one query"
    eng.reset_sequence()
    torch.cuda.synchronize()
    assert int((eng.fsqm["ids"] >= 0).sum()) == 0 and eng.fsqm["pool_hc"].cpu().tolist() == [0, 300, 0]


def test_side_state_on_device_vs_reference_pins():
    """The device side-state path (moy_track_state_update: copy filter + renumbering + FSQM memory in HBM) against the
    REFERENCE's own state over the many-birth stream of tests/golden/state.npz (captured from RuntimeTrackerBase.update's
    return value and FSQM.online_update, head.py:1245-1283, fsqm.py:51-180)."""
    g = golden("state")
    cfg, arch, sd = fixture("tiny")
    sd = dict(sd)
    key = f"model.{len(arch.layers)}.decoder.dec_score_head.{arch.ndl - 1}.bias"
    sd[key] = sd[key] + float(g["bias_shift"])
    eng = TrackEngine(arch, sd, cfg["H"], cfg["W"], batch=1, dtype=torch.float32, side_state=True)
    for t in range(int(g["frames"])):
        out = eng.forward(torch.from_numpy(frames_u8(cfg, t, 1)).to(DEV))
        torch.cuda.synchronize()
        assert np.array_equal(out["obj_idxes"][0].cpu().numpy(), g[f"{t}.obj_idxes"]), t
        n = int(eng.n_copy[0])
        assert eng.copy_ids[0, :n].cpu().tolist() == g[f"{t}.copy_ids"].tolist(), t
        rows = eng.copy_rows[0, :n].cpu().long()
        assert np.allclose(out["boxes"][0].cpu()[rows].numpy(), g[f"{t}.copy_boxes"], atol=2e-4)
        f = {k: v.cpu().numpy() for k, v in eng.fsqm.items()}
        assert np.array_equal(f["ids"], g[f"{t}.fsqm.ids"]) and np.array_equal(f["low"], g[f"{t}.fsqm.low"]), t
        assert np.allclose(f["conf"], g[f"{t}.fsqm.conf"], atol=2e-4) and np.allclose(f["boxes"], g[f"{t}.fsqm.boxes"], atol=2e-4)
        assert np.allclose(f["mem"], g[f"{t}.fsqm.mem"], atol=2e-3)
        head, cnt, ovf = f["pool_hc"]
        assert ovf == 0 and [int(f["pool"][(head + k) % len(f["pool"])]) for k in range(cnt)] == g[f"{t}.fsqm.pool"].tolist()


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_c1_yolov8n_detect_engine(dt):
    """Config C1 on the device: same backbone kernels + Detect decode + NMS + scale_boxes, vs the
    reference rows (fp32: exact row set/order, boxes 1e-2 px) and vs the oracle (bf16: stated bar)."""
    from mo_yolo_amd.config import build_detect_arch
    from mo_yolo_amd.synth import SyntheticSequence
    from mo_yolo_amd.weights import make_fixture_state_dict
    g = golden("c1")
    arch = build_detect_arch()
    sd = make_fixture_state_dict(arch, 5)
    seq = SyntheticSequence(0, 640, 640, "mot17")
    fr = torch.from_numpy(seq.frames(0, 2)).to(DEV)
    for key, hw in (("rows", (640, 640)), ("rows_480x600", (480, 600))):
        eng = TrackEngine(arch, sd, 640, 640, batch=2, dtype=dt, conf=0.25, iou=0.7, max_det=300, orig_hw=hw)
        out = eng.forward(fr)
        torch.cuda.synchronize()
        assert tuple(out["y"].shape) == (2, 84, 8400)
        if dt == torch.float32:
            assert np.allclose(out["y"][0].cpu().numpy().reshape(-1)[g["y0.idx"]], g["y0.val"], atol=2e-3, rtol=1e-4)
        for t in range(2):
            want = g[f"post.{t}.{key}"]
            n = int(out["n_rows"][t])
            got = out["rows"][t, :n].cpu().numpy()
            if dt == torch.float32:
                assert n == len(want)
                assert np.array_equal(got[:, 5], want[:, 5])
                assert np.allclose(got[:, :5], want[:, :5], atol=2e-2, rtol=1e-4)
            else:
                assert abs(n - len(want)) <= max(3, len(want) // 10)

"""GPU parity of the drop-in module surface (mo_yolo_amd.modules / predictor) against the oracle
and the reference goldens: these tests read like the reference's own module calls."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from mo_yolo_amd import modules as M
from mo_yolo_amd.predictor import TrackPredictor
from mo_yolo_amd.synth import SyntheticSequence
from oracle import track_oracle as O
from tests._util import fixture, frames_u8, golden, net_input

DEV = "cuda"


def load(mod, sd, prefix):
    sub = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    mod.load_state_dict(sub, strict=True)
    return mod.to(DEV)


@pytest.mark.parametrize("dt,atol", [(torch.float32, 1e-4), (torch.bfloat16, 6e-2)])
def test_conv_c2f_sppf_modules(dt, atol):
    cfg, arch, sd = fixture("tiny")
    x = net_input(cfg, 0, 2)
    with torch.no_grad():
        _, outs = O.backbone_neck(x, sd, arch, return_all=True)
    specs = {L.i: L for L in arch.layers}
    # layer 1: Conv 3x3 s2 ; layer 2: C2f(shortcut) ; layer 9: SPPF ; layer 12: C2f(no shortcut)
    for i, ctor in ((1, lambda L: M.Conv(L.c1, L.c2, L.k, L.s)), (2, lambda L: M.C2f(L.c1, L.c2, L.n, L.shortcut)),
                    (9, lambda L: M.SPPF(L.c1, L.c2, L.k)), (12, lambda L: M.C2f(L.c1, L.c2, L.n, L.shortcut))):
        Ls = specs[i]
        mod = load(ctor(Ls), sd, f"model.{i}.")
        src = outs[Ls.src[0]] if Ls.kind != "C2f" or i != 12 else outs[11]
        y = mod(src.to(DEV, dt))
        assert y.shape == outs[i].shape
        assert torch.allclose(y.float().cpu(), outs[i], atol=atol), (i, float((y.float().cpu() - outs[i]).abs().max()))
    stem = load(M.Conv(3, specs[0].c2, 3, 2), sd, "model.0.")
    y = stem(x.to(DEV))
    assert torch.allclose(y.float().cpu(), outs[0], atol=1e-4)


def test_decoder_layer_and_msdeformattn_modules_vs_golden():
    """MOTRDecoderLayer.forward / MSDeformAttn.forward with the reference's own call signature,
    checked against tensors the reference produced (tests/golden/tiny.npz)."""
    cfg, arch, sd = fixture("tiny")
    g = golden("tiny")
    d = f"model.{len(arch.layers)}.decoder"
    x = net_input(cfg, 0, 1)
    with torch.no_grad():
        r = O.forward(x, sd, arch)
    layer = load(M.MOTRDecoderLayer(256, 8, 1024, 0.0, None, 3, 4), sd, d + ".decoder.layers.0.")
    embed, qpos = r["embed"].to(DEV), r["query_pos"].to(DEV)
    ref = r["refer_bbox_logit"].sigmoid().to(DEV)
    out = layer(embed, ref, r["feats"].to(DEV), r["shapes"], track_query_pos=qpos)
    assert np.allclose(out.cpu().numpy(), g["t0.dec0.out"], atol=3e-4)
    # cross attention alone (query = n1 + pos as in transformer.py:643)
    n1 = torch.from_numpy(g["t0.dec0.n1"]).to(DEV)
    ca = layer.cross_attn(n1 + qpos, ref.unsqueeze(2), r["feats"].to(DEV), r["shapes"])
    assert np.allclose(ca.cpu().numpy(), g["t0.dec0.ca"], atol=3e-4)
    # the python op seam actually on the path (nn/modules/utils.py:41) with reference-produced operands
    value = torch.from_numpy(g["t0.dec0.value"]).view(1, -1, 8, 32).to(DEV) if "t0.dec0.value" in g else None
    if value is not None:
        o = M.multi_scale_deformable_attn_pytorch(value, r["shapes"], torch.from_numpy(g["t0.dec0.msda_loc"]).to(DEV),
                                                  torch.from_numpy(g["t0.dec0.msda_aw"]).to(DEV))
        assert np.allclose(o.cpu().numpy(), g["t0.dec0.msda_out"], atol=1e-5)


def test_msdeformattn_value_mask_zeroes_the_projected_value():
    """MSDeformAttn.forward(..., value_mask): `value.masked_fill(value_mask[..., None], 0)` AFTER value_proj (transformer.py:264-266) --
    against the same module called on explicitly masked projected values through its sampling core, and against no mask."""
    cfg, arch, sd = fixture("tiny")
    d = f"model.{len(arch.layers)}.decoder"
    with torch.no_grad():
        r = O.forward(net_input(cfg, 0, 1), sd, arch)
    ca = load(M.MOTRDecoderLayer(256, 8, 1024, 0.0, None, 3, 4), sd, d + ".decoder.layers.0.").cross_attn
    q = (r["embed"] + r["query_pos"]).to(DEV)
    ref = r["refer_bbox_logit"].sigmoid().to(DEV).unsqueeze(2)
    feats = r["feats"].to(DEV)
    S = feats.shape[1]
    mask = (torch.arange(S) % 3 == 0)[None]                        # True = zero this token's value
    got = ca(q, ref, feats, r["shapes"], value_mask=mask.to(DEV))
    v2d = ca.value_proj.rows(feats.reshape(-1, 256).contiguous())
    v2d[mask[0].to(DEV)] = 0
    want = ca.output_proj.rows(ca.core(q.reshape(-1, 256).contiguous(), ref.reshape(-1, 1, 4)[:, 0].float().contiguous(), v2d, 1, q.shape[1],
                                       [tuple(s) for s in r["shapes"]])).view_as(got)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    assert not torch.equal(got, ca(q, ref, feats, r["shapes"]))


def test_transformer_decoder_module():
    cfg, arch, sd = fixture("tiny")
    d = f"model.{len(arch.layers)}.decoder"
    with torch.no_grad():
        r = O.forward(net_input(cfg, 0, 1), sd, arch)
    dec = load(M.MYDecoder(nc=arch.nc, ch=arch.head_ch, nq=arch.nq), sd, d + ".")
    boxes, scores, hs = dec.decoder(r["embed"].to(DEV), r["refer_bbox_logit"].to(DEV), r["feats"].to(DEV), r["shapes"],
                                    dec.dec_bbox_head, dec.dec_score_head, dec.query_pos_head,
                                    track_query_embed=r["query_pos"].to(DEV))
    assert boxes.shape == (1, 1, arch.nq, 4) and scores.shape == (1, 1, arch.nq, arch.nc)
    assert torch.allclose(boxes[0].cpu(), r["dec_bboxes"], atol=1e-4)
    assert torch.allclose(scores[0].cpu(), r["dec_scores"], atol=1e-3)
    assert torch.allclose(hs.cpu(), r["hs"], atol=5e-4)


@pytest.mark.parametrize("name", ["tiny", "tiny3"])
def test_tracking_model_predict_and_motrtrack_forward(name):
    """TrackingModel.predict (tasks.py:486) and MOTRTrack.forward (head.py:191) return structures and
    values vs the reference goldens."""
    cfg, arch, sd = fixture(name)
    g = golden(name)
    model = M.TrackingModel(cfg["depth"], cfg["width"], cfg["nc"], cfg["nq"]).load_reference(sd)
    x = net_input(cfg, 0, 1).to(DEV)
    (y, x7), inst = model.predict(x)
    torch.cuda.synchronize()
    assert y.shape == (1, cfg["nq"], 4 + cfg["nc"])
    assert np.allclose(y[0].cpu().numpy(), g["y"][0], atol=2e-4)
    assert len(x7) == 7 and x7[0].shape == (1, 1, cfg["nq"], 4) and x7[4] is None
    assert np.allclose(x7[2].cpu().numpy(), g["t0.enc_bboxes"], atol=1e-4)
    assert np.allclose(x7[3].cpu().numpy(), g["t0.enc_scores"], atol=2e-4)
    assert np.array_equal(inst.obj_idxes.view(-1).cpu().numpy(), g["obj_idxes"][0])
    assert inst.obj_idxes.shape == (cfg["nq"], 1)                       # head.py:166 shape convention
    active = inst[inst.obj_idxes >= 0]
    assert len(active) == int((g["obj_idxes"][0] >= 0).sum())
    # MOTRTrack.forward on the three pyramid levels produced by the oracle backbone
    with torch.no_grad():
        feats = O.backbone_neck(net_input(cfg, 0, 1), sd, arch)
    head = model.model[-1]
    (y2, _), inst2 = head([f.to(DEV) for f in feats])
    assert np.allclose(y2[0].cpu().numpy(), g["y"][0], atol=2e-4)
    assert np.array_equal(inst2.obj_idxes.view(-1).cpu().numpy(), g["obj_idxes"][0])


def test_mydecoder_forward_seven_tuple_and_instances_fields():
    """Boundary row b3: `MYDecoder.forward(x, ...)` returns the reference's 7-tuple (head.py:873-985) and the Instances of
    `MOTRTrack.forward` carry the 11 fields of `_generate_empty_tracks` (head.py:150-189) with the reference's shapes/dtypes;
    returned tensors are fresh per call (a second call must not change what the first returned); loading weights drops the
    cached plan."""
    cfg, arch, sd = fixture("tiny")
    g = golden("tiny")
    nq, nc = cfg["nq"], cfg["nc"]
    model = M.TrackingModel(cfg["depth"], cfg["width"], nc, nq).load_reference(sd)
    head = model.model[-1]
    with torch.no_grad():
        f0 = [f.to(DEV) for f in O.backbone_neck(net_input(cfg, 0, 1), sd, arch)]
        f1 = [f.to(DEV) for f in O.backbone_neck(net_input(cfg, 1, 1), sd, arch)]
    x7 = head.decoder(f0)
    torch.cuda.synchronize()
    assert isinstance(x7, tuple) and len(x7) == 7
    dec_bboxes, dec_scores, enc_bboxes, enc_scores, dn_meta, init_ref, hs = x7
    assert dec_bboxes.shape == (1, 1, nq, 4) and dec_scores.shape == (1, 1, nq, nc) and dn_meta is None
    assert enc_bboxes.shape == (1, nq, 4) and enc_scores.shape == (1, nq, nc) and init_ref.shape == (1, nq, 4) and hs.shape == (1, nq, 256)
    assert np.allclose(dec_bboxes[0, 0].cpu().numpy(), g["y"][0][:, :4], atol=2e-4)
    assert np.allclose(dec_scores[0, 0].sigmoid().cpu().numpy(), g["y"][0][:, 4:], atol=2e-4)
    assert np.allclose(enc_bboxes.cpu().numpy(), g["t0.enc_bboxes"], atol=1e-4)
    assert np.allclose(enc_scores.cpu().numpy(), g["t0.enc_scores"], atol=2e-3, rtol=1e-5)
    assert np.allclose(init_ref.cpu().numpy(), g["t0.enc_bboxes"], atol=1e-4)       # track_ref_pts.sigmoid() of the top-k boxes (head.py:960)
    assert abs(float(hs.double().sum()) - float(g["t0.inst.output_embedding.sum"])) < 5e-2
    with pytest.raises(NotImplementedError):
        head.decoder(f0, track_ref_pts=torch.zeros(3, 4, device=DEV))
    # MOTRTrack.forward: the 11 fields, and no aliasing of engine buffers
    (y_a, x7_a), inst = head(f0)
    want = {"ref_pts": (nq, 4), "query_pos": (nq, 256), "output_embedding": (nq, 256), "obj_idxes": (nq, 1), "matched_gt_idxes": (nq,),
            "disappear_time": (nq, 1), "iou": (nq,), "scores": (nq,), "track_scores": (nq, 4), "pred_boxes": (nq, 4), "pred_logits": (nq, nc)}
    assert set(inst.get_fields()) == set(want)
    for k, shp in want.items():
        assert tuple(inst.get(k).shape) == shp, k
        assert inst.get(k).dtype == (torch.long if k in ("obj_idxes", "matched_gt_idxes", "disappear_time") else torch.float32), k
    assert np.allclose(inst.pred_logits.cpu().numpy(), g["t0.inst.pred_logits"], atol=2e-3)
    keep = {k: v.clone() for k, v in inst.get_fields().items()}
    y_keep = y_a.clone()
    (y_b, _), inst_b = head(f1)                                        # same cached plan, other frame
    torch.cuda.synchronize()
    assert torch.equal(y_a, y_keep) and all(torch.equal(inst.get(k), v) for k, v in keep.items()), "outputs alias engine buffers"
    assert not torch.equal(y_b, y_a)
    assert len(head.decoder._engines) == 1
    head.load_state_dict(head.state_dict())                            # (re)loading weights must drop the cached plan
    assert len(head.decoder._engines) == 0


def test_inference_single_image_surface():
    """Upstream MOTR's per-frame entry (MOTR/models/motr.py:580-598) over the temporal engine: first call with
    track_instances=None, then the returned Instances is handed back; ids persist, boxes come in original-image pixels."""
    cfg, arch, sd = fixture("tiny")
    model = M.TrackingModel(cfg["depth"], cfg["width"], cfg["nc"], cfg["nq"]).load_reference(sd)
    ori = (480, 800)
    ti, seen = None, []
    for t in range(4):
        x = net_input(cfg, t, 1).to(DEV)
        r = model.inference_single_image(x, ori, ti, track_slots=32)
        torch.cuda.synchronize()
        assert set(r) == {"track_instances", "ref_pts"}
        ti = r["track_instances"]
        n = len(ti.obj_idxes)
        assert ti.boxes.shape == (n, 4) and ti.scores.shape == (n,) and ti.labels.shape == (n,) and ti.query_pos.shape == (n, 256)
        if n:
            b = ti.boxes.cpu()
            pb = ti.pred_boxes.cpu()
            assert torch.allclose(b[:, 2] - b[:, 0], pb[:, 2] * ori[1], atol=1e-3) and torch.allclose(b[:, 3] - b[:, 1], pb[:, 3] * ori[0], atol=1e-3)
            assert len(set(ti.obj_idxes.tolist())) == n and int(ti.obj_idxes.min()) >= 0
        assert r["ref_pts"].shape == (32 + cfg["nq"], 2)
        seen.append(set(ti.obj_idxes.tolist()))
    assert seen[0] and seen[0] & seen[1], "tracks born in frame 0 must survive into frame 1 with their ids"
    with pytest.raises(ValueError):
        model.inference_single_image(net_input(cfg, 0, 1).to(DEV), ori, M.Instances((1, 1)))
    r0 = model.inference_single_image(net_input(cfg, 0, 1).to(DEV), ori, None, track_slots=32)    # None restarts the sequence
    assert set(r0["track_instances"].obj_idxes.tolist()) == seen[0]


def test_native_op_module_name_and_autograd_function_as_the_reference_binds_it():
    """`import MultiScaleDeformableAttention as MSDA` (the module name the reference's autograd wrapper imports,
    MOTR/models/ops/functions/ms_deform_attn_func.py:21) resolves to libmoyolo's forward AND backward, and the package's
    `MSDeformAttnFunction` (the counterpart of :24-41 there: same `apply` signature) drives both; checked against the torch
    formulation of the same op (:44-64 there, restated in the oracle)."""
    import MultiScaleDeformableAttention as MSDA
    from mo_yolo_amd import ops as _ops
    assert MSDA.ms_deform_attn_forward is _ops.ms_deform_attn_forward and MSDA.ms_deform_attn_backward is _ops.ms_deform_attn_backward
    MSDeformAttnFunction = M.MSDeformAttnFunction        # calls exactly those two entry points (mo_yolo_amd/modules.py)
    g = torch.Generator().manual_seed(3)
    N, Mh, D, Lq, P = 2, 8, 32, 19, 4
    shapes = torch.tensor([(12, 20), (6, 10), (3, 5)], dtype=torch.long)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    S = int(shapes.prod(1).sum())
    value = (torch.rand(N, S, Mh, D, generator=g) * 0.01)
    loc = torch.rand(N, Lq, Mh, 3, P, 2, generator=g) * 1.2 - 0.1
    aw = torch.rand(N, Lq, Mh, 3, P, generator=g) + 1e-5
    aw = aw / aw.sum(-1, keepdim=True).sum(-2, keepdim=True)
    v_d, l_d, a_d = (t.to(DEV).requires_grad_(True) for t in (value, loc, aw))
    out = MSDeformAttnFunction.apply(v_d, shapes.to(DEV), lsi.to(DEV), l_d, a_d, 64)
    go = torch.rand(out.shape, generator=g) - 0.5
    out.backward(go.to(DEV))
    v_c, l_c, a_c = (t.clone().requires_grad_(True) for t in (value, loc, aw))
    ref = O.msda_core(v_c, [tuple(s) for s in shapes.tolist()], l_c, a_c)
    ref.backward(go)
    assert torch.allclose(out.detach().cpu(), ref.detach(), rtol=1e-2, atol=1e-3)       # the reference's own fp32 bar (ops/test.py:60)
    assert torch.allclose(v_d.grad.cpu(), v_c.grad, rtol=1e-2, atol=1e-5)
    assert torch.allclose(l_d.grad.cpu(), l_c.grad, rtol=1e-2, atol=1e-5)
    assert torch.allclose(a_d.grad.cpu(), a_c.grad, rtol=1e-2, atol=1e-5)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        MSDA.ms_deform_attn_forward(value, shapes, lsi, loc, aw, 64)


@pytest.mark.parametrize("case", ["enc3", "enc4_mask_sigmoid"])
def test_upstream_deformable_encoder_vs_reference_golden(case):
    """DeformableTransformerEncoder / MOTRDeformableTransformerEncoderLayer / upstream MSDeformAttn
    (MOTR/models/deformable_transformer_plus.py:347-415, ops/modules/ms_deform_attn.py:30-121) under the reference's class
    names and state_dict keys, against the reference's own outputs (torch formulation of the op; tests/golden/encoder.npz):
    2-d reference points with valid ratios, position embedding, padding mask, softmax and sigmoid attention."""
    from mo_yolo_amd import motr_upstream as U
    g = golden("encoder")
    nl, nh, npnt, nlayers, sig = (int(v) for v in g[f"{case}.cfg"])
    layer = U.MOTRDeformableTransformerEncoderLayer(256, int(g["d_ffn"]), 0.1, "relu", nl, nh, npnt, sigmoid_attn=bool(sig))
    enc = U.DeformableTransformerEncoder(layer, nlayers)
    sd = {k[len(case) + 4:]: torch.from_numpy(v) for k, v in g.items() if k.startswith(case + ".sd.")}
    enc.load_state_dict(sd, strict=True)                                   # the reference's keys, strictly
    enc = enc.to(DEV)
    shp = torch.from_numpy(g[f"{case}.shapes"])
    lsi = torch.cat((shp.new_zeros((1,)), shp.prod(1).cumsum(0)[:-1]))
    src, pos = torch.from_numpy(g[f"{case}.src"]).to(DEV), torch.from_numpy(g[f"{case}.pos"]).to(DEV)
    vr = torch.from_numpy(g[f"{case}.valid_ratios"]).to(DEV)
    mask = torch.from_numpy(g[f"{case}.mask"]).to(DEV) if f"{case}.mask" in g else None
    y = enc(src, shp.to(DEV), lsi.to(DEV), vr, pos, mask)
    ref_pts = enc.get_reference_points(shp, vr, device=DEV)
    y1 = enc.layers[0](src, pos, ref_pts, shp.to(DEV), lsi.to(DEV), mask)
    torch.cuda.synchronize()
    assert torch.allclose(y1.cpu(), torch.from_numpy(g[f"{case}.layer0_out"]), atol=2e-4, rtol=1e-4)
    assert torch.allclose(y.cpu(), torch.from_numpy(g[f"{case}.out"]), atol=5e-4, rtol=1e-4)
    yb = enc(src.bfloat16(), shp.to(DEV), lsi.to(DEV), vr, pos.bfloat16(), mask)             # 16-bit path: stated bar
    assert float((yb.float().cpu() - torch.from_numpy(g[f"{case}.out"])).abs().max()) < 0.15
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        enc.layers[0].self_attn(src.cpu(), ref_pts.cpu(), src.cpu(), shp, lsi)


def test_qim_update_track_embedding_vs_golden():
    """QueryInteractionModule._update_track_embedding (qim.py:251-301), isolated, vs reference output."""
    _, arch, sd = fixture("tiny")
    g = golden("qim")
    qim = load(M.QueryInteractionModule(None, 256, 256, 512), sd, f"model.{len(arch.layers)}.track_embed.")
    for n in (1, 7, 64):
        inst = M.Instances((1, 1), **{k: torch.from_numpy(g[f"n{n}.in.{k}"]).to(DEV)
                                      for k in ("ref_pts", "output_embedding", "query_pos", "pred_boxes")})
        r = qim._update_track_embedding(inst)
        assert np.allclose(r.query_pos.cpu().numpy(), g[f"n{n}.out.query_pos"], atol=2e-4)
        assert np.allclose(r.ref_pts.cpu().numpy(), g[f"n{n}.out.ref_pts"], atol=1e-5)
    data = {"detect_queries": None, "track_queries": inst}
    assert qim(data) is inst                                            # shipped forward: unchanged (qim.py:340)


@pytest.mark.parametrize("name", ["tiny", "tiny3"])
def test_predictor_stream_rows_and_txt(name):
    cfg, arch, sd = fixture(name)
    g = golden(name)
    T = cfg["frames"]
    pred = TrackPredictor(arch, sd, imgsz=(cfg["H"], cfg["W"]), conf=0.25, batch=2, graph=True)   # ragged tail at T=3
    frames = frames_u8(cfg, 0, T)
    res = pred(list(frames))
    assert len(res) == T
    for t, r in enumerate(res):
        assert np.allclose(r.boxes, g[f"post.{t}.boxes"], atol=5e-2, rtol=1e-5)
        if bool(g[f"post.{t}.is_track"]):
            assert np.array_equal(r.track_id, g[f"post.{t}.track_id"].reshape(-1))
            want = str(g[f"post.{t}.txt"]).strip().split("\n")
            got = r.txt_lines()
            assert [l.split()[:2] for l in got] == [l.split()[:2] for l in want]
            assert np.allclose([[float(v) for v in l.split()[2:]] for l in got],
                               [[float(v) for v in l.split()[2:]] for l in want], atol=2e-4)
    # tensor source: boxes stay normalised (predict.py:66)
    res_t = pred(net_input(cfg, 0, 2))
    for t in range(2):
        assert np.allclose(res_t[t].boxes, g[f"post.{t}.boxes_tensor_src"], atol=2e-4)


def test_predictor_stretch_resizes_foreign_frame_sizes():
    """Frames that are not at network resolution go through LetterBox(scaleFill) first (MOTRtrack/predict.py:96-105):
    device resize == oracle resize (bit-exact input), rows come back in ORIGINAL pixels (predict.py:61-76)."""
    from oracle.preprocess_oracle import resize_linear_u8
    cfg, arch, sd = fixture("tiny")
    H, W = cfg["H"], cfg["W"]
    oh, ow = 90, 150
    # windows of a synthetic scene at another size (pure noise frames carry no active row under the calibrated fixture weights)
    src = SyntheticSequence(6, 128, 192).frames(0, 3)[:, 10:10 + oh, 20:20 + ow].copy()
    pred = TrackPredictor(arch, sd, imgsz=(H, W), conf=0.25, batch=2)
    got = pred(list(src))
    want = pred([resize_linear_u8(f, (H, W)) for f in src])            # network-resolution path, pinned by the goldens above
    assert pred._engines[("u8", (oh, ow))].input.shape == (2, H, W, 3)
    assert sum(len(w) for w in want) > 0
    for r, w in zip(got, want):
        assert r.orig_shape == (oh, ow) and len(r) == len(w)
        scale = np.array([ow / W, oh / H, ow / W, oh / H, 1, 1], np.float32)
        assert np.allclose(r.boxes, w.boxes * scale, atol=1e-3)
        assert (r.track_id is None) == (w.track_id is None)
        if r.track_id is not None:
            assert np.array_equal(r.track_id, w.track_id)
            assert np.allclose([[float(v) for v in l.split()] for l in r.txt_lines()],
                               [[float(v) for v in l.split()] for l in w.txt_lines()], atol=1e-5)


@pytest.mark.parametrize("dt,streams", [(torch.float32, 1), (torch.bfloat16, 1), (torch.bfloat16, 2)])
def test_predictor_host_ring_equals_resident_frames_bit_for_bit(dt, streams):
    """the host-fed pipeline (pageable frames -> pinned ring -> H2D on a copy stream -> input slot j % ring ->
    step -> ONE packed device-to-host copy) returns exactly what the engine computes on frames already resident in HBM: same
    rows, same ids, same counts, bit for bit, over more chunks than the ring is deep and with a ragged tail
    (engine/predictor.py:117-134, 256-344)."""
    from mo_yolo_amd.engine import TrackEngine
    cfg, arch, sd = fixture("tiny")
    Bc, T = 3, 17                                                       # 6 chunks through a ring of 3; the last holds 2 frames
    frames = SyntheticSequence(2, cfg["H"], cfg["W"]).frames(0, T)
    pred = TrackPredictor(arch, sd, imgsz=(cfg["H"], cfg["W"]), conf=0.25, batch=Bc, graph=True, dtype=dt, ring=3, streams=streams)
    # (streams = 2: two engines take the chunks in turn, each on its own HIP stream; results in source order)
    got = pred(frames, paths=[f"f{t}" for t in range(T)])
    assert len(got) == T and [r.path for r in got] == [f"f{t}" for t in range(T)]
    eng = TrackEngine(arch, sd, cfg["H"], cfg["W"], batch=Bc, dtype=dt, orig_hw=(cfg["H"], cfg["W"]))
    dev = torch.from_numpy(frames).cuda()
    n_act = 0
    for s in range(0, T, Bc):
        chunk = dev[s:s + Bc]
        k = chunk.shape[0]
        if k < Bc:
            chunk = torch.cat([chunk, chunk[-1:].expand(Bc - k, *chunk.shape[1:])])
        o = eng.forward(chunk.contiguous())
        torch.cuda.synchronize()
        rows, tid, nr, ni = (o[key].cpu().numpy() for key in ("rows", "track_id", "n_rows", "n_ids"))
        for b in range(k):
            r = got[s + b]
            assert np.array_equal(r.boxes.view(np.uint32), rows[b, :nr[b]].view(np.uint32))         # bit for bit
            assert (r.track_id is None) == (ni[b] < 0)
            if r.track_id is not None:
                assert np.array_equal(r.track_id, tid[b, :ni[b]])
                n_act += len(r.track_id)
    assert n_act > 0
    # a second call reuses the ring (events of the previous call still attached to its buffers)
    again = pred(frames[:7])
    assert all(np.array_equal(a.boxes, b.boxes) for a, b in zip(again, got[:7]))
    # the generator form keeps the pipeline full across arrays (a ragged chunk in mid-stream included): same results, same order
    parts = [frames[:5], (frames[5:11], [f"g{t}" for t in range(5, 11)]), frames[11:]]
    streamed = [r for chunk in pred.stream(parts) for r in chunk]
    assert len(streamed) == T and [r.path for r in streamed[5:11]] == [f"g{t}" for t in range(5, 11)]
    assert all(np.array_equal(a.boxes, b.boxes) and (a.track_id is None) == (b.track_id is None)
               and (a.track_id is None or np.array_equal(a.track_id, b.track_id)) for a, b in zip(streamed, got))
    # frames that already lie in page-locked memory cross the link from where they are (no staging copy): same results
    pinned = pred(torch.from_numpy(frames).pin_memory())
    assert len(pinned) == T and all(np.array_equal(a.boxes, b.boxes) and np.array_equal(a.track_id, b.track_id)
                                    for a, b in zip(pinned, got) if b.track_id is not None)
    # the packed block is what the separate tensors alias: one copy carries all four
    r2, t2, n2, i2 = eng.unpack_result_block(eng.result_block.cpu())
    assert np.array_equal(r2, eng.rows.cpu().numpy()) and np.array_equal(t2, eng.track_id.cpu().numpy())
    assert np.array_equal(n2, eng.n_rows.cpu().numpy()) and np.array_equal(i2, eng.n_ids.cpu().numpy())


def test_box_iou_vs_validator_formula():
    """moy_box_iou vs `_calculate_box_ious` (val.py:517-553) incl. degenerate boxes, ragged frames and empty sides."""
    from mo_yolo_amd import ops
    from mo_yolo_amd.evaluate import similarity_scores
    from oracle import hota_oracle as H
    rng = np.random.default_rng(0)
    gtb, trb = [], []
    for t, (n, k) in enumerate([(5, 7), (1, 1), (0, 3), (4, 0), (9, 2), (3, 3)]):
        a = rng.uniform(0, 900, (n, 2)); b = rng.uniform(0, 900, (k, 2))
        ga = np.concatenate([a, a + rng.uniform(5, 300, (n, 2))], 1).astype(np.float32)
        tb = np.concatenate([b, b + rng.uniform(5, 300, (k, 2))], 1).astype(np.float32)
        if n > 2:
            ga[1, 2:] = ga[1, :2]                                        # zero-area ground-truth box
            tb[:1] = ga[:1] if k else tb[:1]                             # an exact overlap
        gtb.append(ga); trb.append(tb)
    got = similarity_scores(gtb, trb)
    for t in range(len(gtb)):
        want = H.box_ious_xyxy(gtb[t], trb[t])
        assert got[t].shape == want.shape
        assert np.allclose(got[t], want, atol=1e-6), t
    full = ops.box_iou(torch.from_numpy(gtb[0])[None].to(DEV), torch.from_numpy(trb[0])[None].to(DEV))   # no counts
    assert np.allclose(full[0].cpu().numpy(), H.box_ious_xyxy(gtb[0], trb[0]), atol=1e-6)


def test_track_validator_hota_vs_oracle_pipeline():
    """TrackValidator (val.py:185-507 counterpart) on two synthetic sequences: device IoU + product HOTA == the oracle's
    restatement of the reference evaluator fed with the same predictor rows; MOT txt is written per sequence."""
    import tempfile
    from mo_yolo_amd.evaluate import TrackValidator
    from mo_yolo_amd.synth import SyntheticSequence
    from oracle import hota_oracle as H
    from tests._util import hota_of_tracks
    cfg, arch, sd = fixture("tiny")
    pred = TrackPredictor(arch, sd, imgsz=(cfg["H"], cfg["W"]), conf=0.25, batch=2)
    T = 5
    seqs = []
    for sid in (0, 1):
        seq = SyntheticSequence(sid, cfg["H"], cfg["W"], cfg["style"])
        gts = [seq.boxes(t) for t in range(T)]
        seqs.append(dict(name=f"seq{sid}", frames=seq.frames(0, T), gt_boxes=[g[0] for g in gts], gt_ids=[g[1] for g in gts]))
    with tempfile.TemporaryDirectory() as d:
        res = TrackValidator(pred, save_dir=d)(seqs)
        assert set(res) == {"seq0", "seq1", "COMBINED"}
        for sid in (0, 1):
            rows, ids = [], []
            for r in pred(seqs[sid]["frames"]):
                k = 0 if r.track_id is None else min(len(r.track_id), len(r.boxes))
                rows.append(r.boxes[:k, :4]); ids.append(np.asarray(r.track_id[:k] if k else [], np.int64))
            want = hota_of_tracks(cfg, rows, ids, seq_id=sid)
            for key in ("HOTA", "DetA", "AssA", "LocA", "HOTA_TP", "HOTA_FN", "HOTA_FP"):
                assert np.allclose(res[f"seq{sid}"][key], want[key], atol=1e-6), (sid, key)
            if any(len(i) for i in ids):
                assert os.path.getsize(os.path.join(d, f"seq{sid}.txt")) > 0
        assert np.allclose(res["COMBINED"]["HOTA_TP"], res["seq0"]["HOTA_TP"] + res["seq1"]["HOTA_TP"])


def test_batched_frames_equal_single_frames():
    """Batching frames is result-neutral (no cross-frame state in the shipped path)."""
    from mo_yolo_amd.engine import TrackEngine
    cfg, arch, sd = fixture("tiny")
    fr = torch.from_numpy(frames_u8(cfg, 0, 3)).to(DEV)
    eb = TrackEngine(arch, sd, cfg["H"], cfg["W"], batch=3)
    e1 = TrackEngine(arch, sd, cfg["H"], cfg["W"], batch=1)
    ob = {k: v.clone() for k, v in eb.forward(fr).items()}
    for b in range(3):
        o1 = e1.forward(fr[b:b + 1])
        torch.cuda.synchronize()
        assert torch.equal(o1["obj_idxes"][0], ob["obj_idxes"][b])
        assert torch.allclose(o1["y"][0], ob["y"][b], atol=1e-6)


def test_masked_token_hazard_is_flagged():
    """if a masked (+inf anchor) token is selected the reference goes NaN; the engine
    must report it (n_masked > 0) instead of silently continuing."""
    from mo_yolo_amd.engine import TrackEngine
    cfg, arch, sd = fixture("tiny")
    sd = dict(sd)
    d = f"model.{len(arch.layers)}.decoder"
    sd[d + ".enc_output.0.bias"] = sd[d + ".enc_score_head.weight"].mean(0) * 5.0     # masked tokens now score HIGH
    eng = TrackEngine(arch, sd, cfg["H"], cfg["W"], batch=1)
    out = eng.forward(torch.from_numpy(frames_u8(cfg, 0, 1)).to(DEV))
    torch.cuda.synchronize()
    assert int(out["n_masked"][0]) > 0

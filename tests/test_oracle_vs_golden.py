"""Pin the CPU oracle (oracle/track_oracle.py) against golden vectors produced by the imported
reference (tests/golden/make_golden.py).  CPU only; this is what makes the oracle trustworthy as
the parity arbiter for the `-m gpu` tests."""
import numpy as np
import pytest
import torch

from oracle import track_oracle as O
from tests._util import check_close, fixture, golden, net_input, frames_u8

TOL = 2e-5   # fp32 CPU vs fp32 CPU, different op decomposition only
RTOL = 2e-5  # ... relative part: with calibrated BatchNorm statistics (round 3) the activations are O(1-10), not O(0.1)


@pytest.mark.parametrize("name", ["tiny", "tiny3", "c2", "full"])
def test_seams_frame0(name):
    g = golden(name)
    cfg, arch, sd = fixture(name)
    x = net_input(cfg, 0, 1)
    with torch.no_grad():
        feats_in, outs = O.backbone_neck(x, sd, arch, return_all=True)
        for i, o in enumerate(outs):
            check_close(o, g, f"t0.L{i}", atol=TOL, rtol=1e-5)
        trace = {}
        r = O.head_forward(feats_in, sd, arch, trace=trace)
    d = f"model.{len(arch.layers)}.decoder"
    assert [tuple(s) for s in g["shapes"]] == [tuple(s) for s in r["shapes"]]
    assert np.array_equal(g["valid_mask"], r["valid"][0, :, 0].numpy())
    check_close(r["features"], g, "t0.enc_features", atol=TOL, rtol=RTOL)
    check_close(r["enc_scores_all"], g, "t0.enc_scores_all", atol=TOL, rtol=RTOL)
    fin = torch.isfinite(r["anchors"])
    assert torch.equal(r["anchors"][fin], torch.from_numpy(g["t0.anchors"])[fin]) if "t0.anchors" in g else True
    assert np.array_equal(r["topk_ind"].numpy().reshape(-1), g["t0.topk_ind"].reshape(-1)), "top-k order"
    check_close(r["embed"], g, "t0.embed0", atol=TOL, rtol=RTOL)
    check_close(r["refer_bbox_logit"], g, "t0.refer_bbox_logit", atol=TOL, rtol=RTOL)
    check_close(r["query_pos"], g, "t0.query_pos", atol=5e-5)
    check_close(r["enc_bboxes"], g, "t0.enc_bboxes", atol=TOL, rtol=RTOL)
    check_close(r["enc_scores"], g, "t0.enc_scores", atol=TOL, rtol=RTOL)
    for li in range(arch.ndl):
        t = trace[li]
        check_close(t["sa"].transpose(0, 1), g, f"t0.dec{li}.sa", atol=TOL, rtol=RTOL)   # reference MHA is [L, B, E]
        check_close(t["n1"], g, f"t0.dec{li}.n1", atol=TOL, rtol=RTOL)
        check_close(t["msda_loc"], g, f"t0.dec{li}.msda_loc", atol=TOL, rtol=RTOL)
        check_close(t["msda_aw"], g, f"t0.dec{li}.msda_aw", atol=TOL, rtol=RTOL)
        check_close(t["msda_core"], g, f"t0.dec{li}.msda_out", atol=TOL, rtol=RTOL)
        check_close(t["ca"], g, f"t0.dec{li}.ca", atol=TOL, rtol=RTOL)
        check_close(t["n2"], g, f"t0.dec{li}.n2", atol=TOL, rtol=RTOL)
        check_close(t["out"], g, f"t0.dec{li}.out", atol=TOL, rtol=RTOL)
        check_close(t["bbox_delta"], g, f"t0.dec{li}.bbox_delta", atol=TOL, rtol=RTOL)
    assert torch.allclose(r["y"][0], torch.from_numpy(g["y"][0]), atol=TOL, rtol=RTOL)


@pytest.mark.parametrize("name", ["tiny", "tiny3", "c2", "full"])
def test_stream_y_and_ids(name):
    """Per-frame reset semantics: ids restart at 0 in every frame, in query order."""
    g = golden(name)
    cfg, arch, sd = fixture(name)
    nfr = cfg["frames"] if name != "c2" else 3
    for t in range(nfr):
        with torch.no_grad():
            r = O.forward(net_input(cfg, t, 1), sd, arch)
        y = r["y"][0]
        assert torch.allclose(y, torch.from_numpy(g["y"][t]), atol=5e-5), t
        scores = r["dec_scores"][0].sigmoid().max(-1).values
        assert torch.allclose(scores, torch.from_numpy(g["scores"][t]), atol=5e-5)
        ids = O.assign_ids(scores)
        assert np.array_equal(ids.numpy(), g["obj_idxes"][t]), (t, ids, g["obj_idxes"][t])
        ids2, _, _ = O.assign_ids_loop(scores)
        assert ids2 == ids.tolist()
        # predictor rows (a20)
        fr = frames_u8(cfg, t, 1)
        rows, tid = O.postprocess(y, r["dec_scores"][0], ids, conf=0.25, orig_hw=fr.shape[1:3])
        assert bool(g[f"post.{t}.is_track"]) == (tid is not None)
        assert np.allclose(rows.numpy(), g[f"post.{t}.boxes"], atol=2e-2, rtol=1e-5)   # pixels
        if tid is not None:
            assert np.array_equal(tid.numpy(), g[f"post.{t}.track_id"].reshape(-1))
            lines = O.txt_lines(rows, tid, fr.shape[1:3])
            want = str(g[f"post.{t}.txt"]).strip().split("\n")
            assert len(lines) == len(want)
            for a, b in zip(lines, want):
                fa, fb = a.split(), b.split()
                assert fa[:2] == fb[:2]
                assert np.allclose([float(v) for v in fa[2:]], [float(v) for v in fb[2:]], atol=2e-5)
        rows_n, _ = O.postprocess(y, r["dec_scores"][0], ids, conf=0.25, orig_hw=None)
        assert np.allclose(rows_n.numpy(), g[f"post.{t}.boxes_tensor_src"], atol=5e-5)


@pytest.mark.slow
def test_c4_frame0():
    g = golden("c4")
    cfg, arch, sd = fixture("c4")
    with torch.no_grad():
        r = O.forward(net_input(cfg, 0, 1), sd, arch)
    assert np.array_equal(r["topk_ind"].numpy().reshape(-1), g["t0.topk_ind"].reshape(-1))
    assert torch.allclose(r["y"][0], torch.from_numpy(g["y"][0]), atol=5e-5)
    ids = O.assign_ids(r["dec_scores"][0].sigmoid().max(-1).values)
    assert np.array_equal(ids.numpy(), g["obj_idxes"][0])


def test_msda_core_kats():
    """MSDA core vs the reference op (grid_sample formulation) incl. zero-padding edge taps;
    construction follows MOTR/models/ops/test.py:21-30."""
    g = golden("msda_kat")
    for name in ("kat_tiny", "kat_heads8", "kat_odd"):
        v = torch.from_numpy(g[name + ".value"]); loc = torch.from_numpy(g[name + ".loc"])
        aw = torch.from_numpy(g[name + ".aw"]); shapes = [tuple(s) for s in g[name + ".shapes"]]
        o = O.msda_core(v, shapes, loc, aw)
        assert torch.allclose(o, torch.from_numpy(g[name + ".out"]), atol=1e-7, rtol=1e-5), name
        o64 = O.msda_core(v.double(), shapes, loc.double(), aw.double())
        assert torch.allclose(o64, torch.from_numpy(g[name + ".out_f64"]), atol=1e-12, rtol=1e-9), name


def test_msda_backward_kats():
    """Oracle backward (restating ms_deform_im2col_cuda.cuh:301-400) vs autograd through the reference's torch op
    (the comparison its ops/test.py:66-86 makes with gradcheck)."""
    g = golden("msda_kat")
    for name in ("kat_tiny", "kat_heads8", "kat_odd"):
        shapes = [tuple(s) for s in g[name + ".shapes"]]
        for tag, dt, tol in (("", torch.float32, 2e-7), ("_f64", torch.float64, 1e-15)):
            v, loc, aw, go = (torch.from_numpy(g[f"{name}.{k}"]).to(dt) for k in ("value", "loc", "aw", "gout"))
            for k, t in zip(("gvalue", "gloc", "gaw"), O.msda_core_backward(v, shapes, loc, aw, go)):
                assert torch.allclose(t, torch.from_numpy(g[f"{name}.{k}{tag}"]), atol=tol, rtol=1e-5), (name, k, tag)


def test_qim_isolated():
    g = golden("qim")
    _, arch, sd = fixture("tiny")
    t = f"model.{len(arch.layers)}.track_embed"
    for n in (1, 7, 64):
        i = {k: torch.from_numpy(g[f"n{n}.in.{k}"]) for k in ("ref_pts", "output_embedding", "query_pos", "pred_boxes")}
        qp, rp = O.qim_update_track_embedding(i["ref_pts"], i["output_embedding"], i["query_pos"], i["pred_boxes"], sd, t)
        assert torch.allclose(qp, torch.from_numpy(g[f"n{n}.out.query_pos"]), atol=2e-5)
        assert torch.allclose(rp, torch.from_numpy(g[f"n{n}.out.ref_pts"]), atol=1e-6)


def test_filter_and_copy_semantics():
    b = np.array([[.5, .5, .2, .2], [.5, .5, .2, .21], [.1, .1, .05, .05], [.5, .5, .2, .2]], np.float32)
    assert O.filter_tracks(b) == [True, False, True, False]
    rows, ids = O.tracker_update_copy([.9, .9, .9, .9], b, [0, 1, 2, 3])
    assert rows == [0, 2] and ids == [0, 1]


def test_hota_restatement_vs_reference_evaluator():
    """oracle/hota_oracle.py vs HOTA.eval_sequence of the reference (utils/hota.py:24-164) on the
    validator's input layout, incl. the evaluator's in-place id shifting."""
    from oracle import hota_oracle as H
    g = golden("hota")
    for case in ("easy", "hard", "sparse", "single"):      # sparse / single: the K == 1, K == 0 and n == 1 branches
        T = int(g[f"{case}.T"])
        r = H.eval_sequence([g[f"{case}.gt_ids.{t}"] for t in range(T)], [g[f"{case}.tracker_ids.{t}"] for t in range(T)],
                            [g[f"{case}.sim.{t}"] for t in range(T)], int(g[f"{case}.num_gt_ids"]), int(g[f"{case}.num_tracker_ids"]))
        for k in ("HOTA", "DetA", "AssA", "DetRe", "DetPr", "AssRe", "AssPr", "LocA", "OWTA", "HOTA_TP", "HOTA_FN", "HOTA_FP"):
            assert np.allclose(r[k], g[f"{case}.res.{k}"], atol=1e-7), (case, k)


def test_c1_yolov8n_detect_vs_reference():
    """Config C1 (YOLOv8n detect, 640x640): Detect decode, NMS and scale_boxes restatements vs the
    reference DetectionModel + DetectionPredictor.postprocess outputs (tests/golden/c1.npz)."""
    from mo_yolo_amd.config import build_detect_arch
    from mo_yolo_amd.synth import SyntheticSequence, to_network_input
    from mo_yolo_amd.weights import make_fixture_state_dict, state_dict_digest
    g = golden("c1")
    arch = build_detect_arch()
    sd = make_fixture_state_dict(arch, 5)
    assert state_dict_digest(sd) == str(g["weights_sha256"])
    seq = SyntheticSequence(0, 640, 640, "mot17")
    for t in range(2):
        with torch.no_grad():
            y = O.detect_forward(to_network_input(seq.frames(t, 1)), sd, arch)
        assert tuple(y.shape) == (1, 84, 8400)
        if t == 0:
            assert np.allclose(y[0].numpy().reshape(-1)[g["y0.idx"]], g["y0.val"], atol=5e-4, rtol=1e-5)
        rows = O.non_max_suppression(y, 0.25, 0.7, 300)[0]
        for key, hw in ((f"post.{t}.rows", (640, 640)), (f"post.{t}.rows_480x600", (480, 600))):
            r = rows.clone()
            r[:, :4] = O.scale_boxes((640, 640), r[:, :4], hw)
            assert r.shape == g[key].shape
            assert np.allclose(r.numpy(), g[key], atol=2e-3, rtol=1e-5)


def test_resize_oracle_known_answers():
    """cv2 is absent (parity unpinned, see oracle/preprocess_oracle.py): known answers of OpenCV's 8-bit INTER_LINEAR."""
    from oracle.preprocess_oracle import letterbox_scalefill, resize_linear_u8
    assert resize_linear_u8(np.array([[[0], [255]]], np.uint8), (1, 4))[0, :, 0].tolist() == [0, 64, 191, 255]
    r = np.random.default_rng(0).integers(0, 256, (14, 18, 3), dtype=np.uint8)
    assert np.array_equal(resize_linear_u8(r, (14, 18)), r)                           # unit scale: taps (2048, 0)
    assert letterbox_scalefill(r, (14, 18)) is r                                       # augment.py:586 no-op
    assert (resize_linear_u8(np.full((5, 7, 3), 77, np.uint8), (32, 64)) == 77).all()  # constants survive the fixed point
    area = resize_linear_u8(r, (7, 9))                                                 # exact 2x shrink -> INTER_AREA
    v = r.astype(np.int64)
    assert np.array_equal(area, ((v[0::2, 0::2] + v[0::2, 1::2] + v[1::2, 0::2] + v[1::2, 1::2] + 2) >> 2))
    up = resize_linear_u8(r, (33, 40))
    assert up.shape == (33, 40, 3) and up.min() >= r.min() and up.max() <= r.max()     # convex taps stay in range


def test_state_machine_copy_half_and_fsqm_vs_reference_pins():
    """Rows a16 (copy half) + a17 pinned by the reference itself (tests/golden/state.npz, make_golden.py: dump_state):
    `tracker_update_copy` against what RuntimeTrackerBase.update returned (head.py:1245-1283, greedy IoU filter
    :1155-1196 incl. clustered boxes that it really suppresses), `FSQMOracle` against the reference FSQM's memory after every
    frame of a many-birth stream (fsqm.py:51-180; the id pool that grows by recycled -1 ids included)."""
    g = golden("state")
    for c in range(int(g["unit_cases"])):
        sc, bx = g[f"unit{c}.scores"], g[f"unit{c}.boxes"]
        ids, _, nxt = O.assign_ids_loop(torch.from_numpy(sc))
        assert ids == g[f"unit{c}.obj_idxes"].tolist()
        rows, nid = O.tracker_update_copy(sc.tolist(), bx, ids)
        assert nid == g[f"unit{c}.copy_ids"].tolist(), c
        assert np.array_equal(bx[rows], g[f"unit{c}.copy_boxes"])
        assert (max(nid) + 1 if nid else nxt) == int(g[f"unit{c}.max_obj_id"])
    fs = O.FSQMOracle(300, 256)
    used = 0
    for t in range(int(g["frames"])):
        sc, bx, ids, hs = g[f"{t}.scores"], g[f"{t}.boxes"], g[f"{t}.obj_idxes"], g[f"{t}.hs"]
        assert O.assign_ids(torch.from_numpy(sc)).tolist() == ids.tolist()
        rows, nid = O.tracker_update_copy(sc.tolist(), bx, ids.tolist())
        assert nid == g[f"{t}.copy_ids"].tolist() and np.array_equal(bx[rows], g[f"{t}.copy_boxes"])
        det = (sc[rows], bx[rows], hs[rows]) if rows else (sc, bx, hs)      # no active row: FSQM gets the full Instances
        fs.online_update(det[0], det[1], det[2], sc, bx, ids)
        assert np.array_equal(fs.ids, g[f"{t}.fsqm.ids"]) and np.array_equal(fs.low, g[f"{t}.fsqm.low"]), t
        assert np.array_equal(fs.conf, g[f"{t}.fsqm.conf"]) and np.array_equal(fs.boxes, g[f"{t}.fsqm.boxes"])
        assert np.array_equal(fs.mem, g[f"{t}.fsqm.mem"])
        assert fs.pool == g[f"{t}.fsqm.pool"].tolist()
        used = int((fs.ids >= 0).sum())
    assert used > 20, "the stream must really fill the memory"

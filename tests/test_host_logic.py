"""CPU tests of host-side logic: architecture/state_dict surface, sharding over gloo (world 2),
result formatting, synthetic streams."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from mo_yolo_amd.config import build_arch, level_shapes, param_shapes
from mo_yolo_amd.synth import SyntheticSequence, to_network_input
from tests._util import golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_arch_matches_survey_tables():
    a = build_arch(0.33, 0.50, 1, 300)
    assert a.head_ch == (128, 256, 256)                        # SURVEY §8 "Head input channels at s-scale"
    assert [L.n for L in a.layers if L.kind == "C2f"] == [1, 2, 2, 1, 1, 1, 1, 1]
    n_params = sum(int(np.prod(s)) for k, s in param_shapes(a).items() if not k.endswith("num_batches_tracked")
                   and "running_" not in k)
    assert n_params == 12849283                                 # 12.85 M params
    assert sum(h * w for h, w in level_shapes(608, 1088)) == 13566
    assert sum(h * w for h, w in level_shapes(1088, 1920)) == 42840


def test_module_state_dict_keys_match_reference_names():
    from mo_yolo_amd.modules import TrackingModel
    m = TrackingModel(0.33, 0.25, nc=3, nq=20)
    want = param_shapes(build_arch(0.33, 0.25, 3, 20))
    got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert set(got) == set(want)
    assert all(got[k] == tuple(want[k]) for k in want)


def test_synthetic_stream_is_deterministic_and_matches_golden_digest():
    import hashlib
    g = golden("c2")
    fr = SyntheticSequence(0, 608, 1088, "mot17").frames(0, 1)
    assert hashlib.sha256(fr.tobytes()).hexdigest() == str(g["frame0_sha256"])
    x = to_network_input(fr)
    assert x.shape == (1, 3, 608, 1088) and float(x.max()) <= 1.0
    assert np.array_equal(fr[0, :, :, ::-1].transpose(2, 0, 1), (x[0] * 255).round().numpy().astype(np.uint8))


def test_track_results_txt_matches_reference_lines():
    from mo_yolo_amd.predictor import TrackResults
    g = golden("tiny")
    for t in range(3):
        r = TrackResults(g[f"post.{t}.boxes"], g[f"post.{t}.track_id"].reshape(-1), (96, 160))
        want = str(g[f"post.{t}.txt"]).strip().split("\n")
        got = r.txt_lines()
        assert len(got) == len(want)
        for a, b in zip(got, want):
            fa, fb = a.split(), b.split()
            assert fa[:2] == fb[:2] and np.allclose([float(v) for v in fa[2:]], [float(v) for v in fb[2:]], atol=2e-6)
        wc = str(g[f"post.{t}.txt_conf"]).strip().split("\n")
        assert [l.split()[-1] for l in r.txt_lines(save_conf=True)] == [l.split()[-1] for l in wc]


_WORKER = r"""
import os, sys, json
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
from mo_yolo_amd import shard
dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
rank, world = dist.get_rank(), dist.get_world_size()
seqs = shard.sequences_for_rank(8, rank, world)
dt = shard.max_over_ranks(1.0 + rank)                 # rank 1 is the slow one
res = shard.gather_objects({{"rank": rank, "seqs": seqs}})
dist.barrier()
if rank == 0:
    print(json.dumps({{"dt": dt, "fps": shard.whole_job_fps(100, dt, world), "res": res}}))
dist.destroy_process_group()
"""


def test_sharding_two_process_gloo(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "w.py"
    script.write_text(_WORKER.format(root=ROOT))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = [p.communicate(timeout=120) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    import json
    d = json.loads(outs[0][0].decode().strip().splitlines()[-1])
    assert d["dt"] == 2.0 and d["fps"] == 100.0                 # MAX over ranks; whole-job aggregate
    assert d["res"][0]["seqs"] == [0, 2, 4, 6] and d["res"][1]["seqs"] == [1, 3, 5, 7]


def test_bench_control_flow_two_ranks_gloo_dry_run():
    """The EXACT multi-GPU code of bench.py -- device pinning per local rank, rank -> sequence map (shard.py), barrier, MAX
    over ranks, rank-0 JSON line -- under torch.distributed.run with 2 ranks on gloo, no GPU (`--dry-run`).  The 8-GPU
    curve itself is the driver's to measure (sequences are independent, no collective on the data path)."""
    import json
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1",
           "--dry-run", "--backend", "gloo", "--config", "c5", "--batch", "8"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines                               # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["scaling"] == "weak" and d["config"]["dry_run"] is True
    assert d["config"]["sequences_of_rank0"] == [0, 2, 4, 6]     # C5: 4 sequences per GPU, sequence i -> rank i mod N
    assert d["config"]["control_backend"] == "gloo"
    # value = frames of ALL ranks / MAX over ranks of the timed region: the slower synthetic rank (1.1x) sets the time
    assert abs(d["value"] - 8 * 5 * 2 / (d["ms_per_step"] * 5 * 1e-3)) < 1e-4 * d["value"]      # (ms_per_step is rounded to 4 digits)
    assert d["ms_per_step"] >= 2.2 * 0.99


def test_bench_control_flow_c3_eight_ranks_gloo_dry_run():
    """Config C3 as the driver launches it (`--gpus 8`, torch.distributed.run, 8 ranks) with the data path stubbed out: the
    exact 8-rank control flow -- rendezvous on 127.0.0.1, one device per local rank (HIP_VISIBLE_DEVICES 0..7), sequence i on
    rank i, barrier, MAX over ranks, one JSON line from rank 0 -- has run somewhere before the first real node sees it."""
    import json
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "4", "--warmup", "1",
           "--dry-run", "--backend", "gloo"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "HIP_VISIBLE_DEVICES", "LOCAL_WORLD_SIZE")}
    env["OMP_NUM_THREADS"] = "1"
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["config"]["frames_per_step_per_gpu"] == 576
    rm = sorted(d["config"]["rank_map"], key=lambda e: e["rank"])
    assert [e["rank"] for e in rm] == list(range(8))
    assert [e["hip_visible_devices"] for e in rm] == [str(i) for i in range(8)]      # one GPU per rank, pinned before HIP starts
    assert [e["sequences"] for e in rm] == [[i] for i in range(8)]                   # sequence i -> rank i, nothing shared
    assert abs(d["value"] - 576 * 4 * 8 / (d["ms_per_step"] * 4 * 1e-3)) < 1e-4 * d["value"]
    # the per-rank table that makes the first hardware run diagnosable -- 8 distinct devices, 8 sequences, 8 timings
    rk = d["config"]["ranks"]
    assert [e["rank"] for e in rk] == list(range(8)) and [e["local_rank"] for e in rk] == list(range(8))
    assert len({e["hip_visible_devices"] for e in rk}) == 8
    assert sorted(sq for e in rk for sq in e["sequences"]) == list(range(8))
    assert all(e["dt_local_s"] > 0 and e["fps_local"] > 0 for e in rk)
    assert max(e["dt_local_s"] for e in rk) <= d["ms_per_step"] * 4 * 1e-3 * 1.0001       # the line's time is the MAX over ranks
    cpus = [e["cpus"] for e in rk]
    assert all(c for c in cpus)
    if len(os.sched_getaffinity(0)) >= 16:
        assert len(set(cpus)) == 8                                # every rank pinned its own slice of the host cores


def test_bench_pins_one_device_per_local_rank(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.delenv("HIP_VISIBLE_DEVICES", raising=False)
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES", raising=False)
    assert bench.pin_device(3) == "3"
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "4,5,6,7")         # a driver that hands this job a subset of the node
    assert bench.pin_device(2) == "6"
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "5")               # already pinned by the launcher
    assert bench.pin_device(0) == "5"
    # a list shorter than the local rank count would put every rank on one GPU -> refuse, unless it is the rehearsal
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "4")
    with pytest.raises(SystemExit):
        bench.pin_device(1)
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "5")
    assert bench.pin_device(1, rehearse=True) == "5"
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "4,5,6,7")
    assert bench.pin_device(3) == "7"                            # never indexes outside the inherited list
    monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
    monkeypatch.delenv("HIP_VISIBLE_DEVICES", raising=False)
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "4,5,6,7")        # ROCr renumbers its subset from 0: HIP index = local rank
    assert bench.pin_device(2) == "2"


def _hota_case(g, case):
    T = int(g[f"{case}.T"])
    return {"num_timesteps": T, "num_gt_ids": int(g[f"{case}.num_gt_ids"]), "num_tracker_ids": int(g[f"{case}.num_tracker_ids"]),
            "gt_ids": [g[f"{case}.gt_ids.{t}"] for t in range(T)], "tracker_ids": [g[f"{case}.tracker_ids.{t}"] for t in range(T)],
            "similarity_scores": [g[f"{case}.sim.{t}"] for t in range(T)]}


def test_product_hota_compat_vs_reference_evaluator():
    """mo_yolo_amd.evaluate.HOTA(compat=True) against outputs of the reference's own evaluator (tests/golden/hota.npz,
    utils/hota.py:24-164), same `eval_sequence(data)` dictionary; the caller's id arrays are left untouched."""
    from mo_yolo_amd.evaluate import HOTA
    g = golden("hota")
    for case in ("easy", "hard", "sparse", "single"):
        data = _hota_case(g, case)
        before = [t.copy() for t in data["tracker_ids"]]
        r = HOTA().eval_sequence(data)
        for k in HOTA.float_array_fields + HOTA.integer_array_fields:
            assert np.allclose(r[k], g[f"{case}.res.{k}"], atol=1e-7), (case, k)
        for k in HOTA.float_fields:
            assert np.isclose(r[k], float(g[f"{case}.res.{k}"]), atol=1e-7), (case, k)
        assert all(np.array_equal(a, b) for a, b in zip(before, data["tracker_ids"]))


def test_product_hota_standard_definition_properties():
    """HOTA(compat=False), the published definition: perfect tracks score 1 at every alpha, relabelling tracker ids is
    neutral, an id switch lowers AssA but not DetA, and combine_sequences weights by TP (hota.py:166-176)."""
    from mo_yolo_amd.evaluate import HOTA, build_hota_data
    T, n = 12, 4
    gt = [np.arange(n) for _ in range(T)]
    eye = [np.eye(n) for _ in range(T)]
    h = HOTA(compat=False)
    perfect = h.eval_sequence(build_hota_data(gt, gt, eye))
    assert np.allclose(perfect["HOTA"], 1) and np.allclose(perfect["AssA"], 1) and np.allclose(perfect["DetA"], 1)
    relabelled = h.eval_sequence(build_hota_data(gt, [np.array([7, 3, 11, 5])[g] for g in gt], eye))
    assert np.allclose(relabelled["HOTA"], 1)
    switched = [g.copy() for g in gt]
    for t in range(T // 2, T):
        switched[t][[0, 1]] = switched[t][[1, 0]]                   # tracks 0 and 1 swap identities half-way
    sw = h.eval_sequence(build_hota_data(gt, switched, eye))
    assert np.allclose(sw["DetA"], 1) and np.allclose(sw["AssA"], 2 / 3)   # half the TPs: 6 / (12 + 12 - 6)
    comb = h.combine_sequences({"a": perfect, "b": sw})
    assert np.allclose(comb["HOTA_TP"], 2 * T * n) and np.allclose(comb["AssA"], (1 + 2 / 3) / 2)
    empty = h.eval_sequence(build_hota_data(gt, [np.zeros(0, np.int64)] * T, [np.zeros((n, 0))] * T))
    assert np.allclose(empty["HOTA_FN"], T * n) and empty["HOTA(0)"] == 0


def test_valid_token_rectangles_of_the_reference_anchor_mask():
    """Round 3 / 4 host logic behind the score pass over the valid tokens and the folded head: the valid mask of
    `_generate_anchors` (head.py:993-1010, axes swapped as shipped) is ONE rectangle per pyramid level at every resolution the
    configs use -- checked against the oracle's restatement of the reference formula -- and anything else is refused."""
    from mo_yolo_amd.engine import _generate_anchors, valid_rectangles
    from oracle.track_oracle import generate_anchors
    for H, W in ((608, 1088), (1088, 1920), (128, 192), (640, 640)):
        shapes = [(H // s, W // s) for s in (8, 16, 32)]
        _, v_ref = generate_anchors(shapes)
        _, v = _generate_anchors(shapes)
        assert torch.equal(v.bool(), v_ref.bool())
        rects = valid_rectangles(v[0, :, 0], shapes)
        assert rects is not None and len(rects) == 3
        off, n = 0, 0
        for (h, w), r in zip(shapes, rects):
            m = v[0, off:off + h * w, 0].bool().view(h, w)
            off += h * w
            if r is None:
                assert not bool(m.any())
                continue
            y0, y1, x0, x1 = r
            assert bool(m[y0:y1 + 1, x0:x1 + 1].all())
            n += (y1 - y0 + 1) * (x1 - x0 + 1)
        assert n == int(v.sum())
    # the swapped axes cut the 1088-wide levels at x / h < 0.99: 54 % of the tokens are valid at 608 x 1088
    shapes = [(76, 136), (38, 68), (19, 34)]
    _, v = _generate_anchors(shapes)
    assert int(v.sum()) == 7317 and valid_rectangles(v[0, :, 0], shapes) == [(1, 75, 1, 74), (1, 37, 0, 37), (0, 18, 0, 18)]
    holed = v[0, :, 0].clone().bool()
    holed[5 * 136 + 7] = False                                   # (row 5, column 7 of the P3 level)
    assert bool(v[0, 5 * 136 + 7, 0])                                   # a hole: not a rectangle any more
    assert valid_rectangles(holed, shapes) is None
    assert valid_rectangles(torch.zeros(13566, dtype=torch.bool), shapes) == [None, None, None]


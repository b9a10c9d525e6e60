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
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["config"]["frames_per_step_per_gpu"] == 1152
    rm = sorted(d["config"]["rank_map"], key=lambda e: e["rank"])
    assert [e["rank"] for e in rm] == list(range(8))
    assert [e["hip_visible_devices"] for e in rm] == [str(i) for i in range(8)]      # one GPU per rank, pinned before HIP starts
    # round 6: the default 1152-frame batch is cut from TWO sequences per GPU (every frame inside its 600-frame sequence):
    # sequence i -> rank i mod 8, nothing shared
    assert [e["sequences"] for e in rm] == [[i, i + 8] for i in range(8)]
    assert abs(d["value"] - 1152 * 4 * 8 / (d["ms_per_step"] * 4 * 1e-3)) < 1e-4 * d["value"]
    # the per-rank table that makes the first hardware run diagnosable -- 8 distinct devices, 8 sequences, 8 timings
    assert len(lines[0]) < 4096
    assert [e["rank"] for e in d["config"]["ranks"]] == list(range(8))          # the line carries the short per-rank records ...
    rk = json.load(open(d["full"]))["config"]["ranks"]                           # ... the side file it names the full ones
    assert [e["rank"] for e in rk] == list(range(8)) and [e["local_rank"] for e in rk] == list(range(8))
    assert len({e["hip_visible_devices"] for e in rk}) == 8
    assert sorted(sq for e in rk for sq in e["sequences"]) == list(range(16))
    assert all(e["dt_local_s"] > 0 and e["fps_local"] > 0 and e["startup_s"] > 0 for e in rk)
    assert max(e["dt_local_s"] for e in rk) <= d["ms_per_step"] * 4 * 1e-3 * 1.0001       # the line's time is the MAX over ranks
    cpus = [e["cpus"] for e in rk]
    assert all(c for c in cpus)
    if len(os.sched_getaffinity(0)) >= 16:
        assert len(set(cpus)) == 8                                # every rank pinned its own slice of the host cores


def _bench(*flags, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "HIP_VISIBLE_DEVICES", "LOCAL_WORLD_SIZE",
                                                            "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          timeout=timeout)


@pytest.mark.parametrize("n", [2, 8])
def test_bench_gpus_n_self_launches_n_ranks_without_torchrun(n, tmp_path):
    """`python bench.py --gpus N` launched plainly means N GPUs -- the parent (which never touches HIP) starts N ranks of
    itself with the environment torch.distributed.run would give them and forwards rank 0's line (ultralytics/utils/dist.py:49-60 is
    the reference's own self-launch)."""
    import json
    p = _bench("--gpus", str(n), "--steps", "3", "--warmup", "1", "--dry-run", "--backend", "gloo", "--full-out", str(tmp_path / "full.json"))
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 4096, (len(lines), [len(l) for l in lines])
    d = json.loads(lines[0])
    assert d["n_gpus"] == n and d["config"]["dry_run"] is True
    rk = d["config"]["ranks"]
    assert [e["rank"] for e in rk] == list(range(n)) and [e["hip_visible_devices"] for e in rk] == [str(i) for i in range(n)]
    assert sorted(sq for e in rk for sq in e["sequences"]) == list(range(2 * n))
    assert abs(d["value"] - 1152 * 3 * n / (d["ms_per_step"] * 3 * 1e-3)) < 1e-4 * d["value"]
    full = json.load(open(tmp_path / "full.json"))                 # the side file the line names holds the same headline
    assert d["full"] == str(tmp_path / "full.json") and full["value"] == d["value"] and full["n_gpus"] == n


def _pids_with_env(tag):
    """PIDs of live processes whose ENVIRONMENT carries MOY_TEST_TAG=<tag> (the launcher's ranks inherit it) -- never a match on
    command lines, which would also find whatever shell happens to quote the pattern."""
    out = []
    needle = f"MOY_TEST_TAG={tag}".encode()
    for d in os.listdir("/proc"):
        if d.isdigit():
            try:
                with open(f"/proc/{d}/environ", "rb") as f:
                    if needle in f.read().split(b"\0"):
                        with open(f"/proc/{d}/stat") as g:
                            if g.read().rsplit(")", 1)[1].split()[0] != "Z":
                                out.append(int(d))
            except OSError:
                pass
    return out


def test_bench_self_launch_tears_the_job_down_when_a_rank_dies_at_start_up():
    """a rank that exits before the rendezvous (a HIP_VISIBLE_DEVICES entry that does not exist, ...) must not leave rank 0
    waiting in init_process_group for its timeout with the launcher blocked behind it: the launcher polls its ranks, terminates the
    others on the first non-zero exit and returns that code -- in seconds, with no rank left running."""
    import time
    tag = f"fail{os.getpid()}"
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "HIP_VISIBLE_DEVICES", "LOCAL_WORLD_SIZE",
                                                            "MASTER_ADDR", "MASTER_PORT")}
    env.update(OMP_NUM_THREADS="1", MOY_TEST_TAG=tag)
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "3", "--warmup", "1", "--dry-run", "--backend", "gloo",
                        "--fail-rank", "2"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode == 7, (p.returncode, p.stderr.decode()[-1500:])
    assert time.time() - t0 < 60
    assert b"rank 2 failed first" in p.stderr and not [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    for _ in range(50):
        left = _pids_with_env(tag)
        if not left:
            break
        time.sleep(0.1)
    assert not left, left


def test_bench_self_launch_kills_its_ranks_when_it_is_terminated():
    """... and a launcher that is itself killed (the driver's `timeout -k`) takes its ranks with it instead of leaving them on the GPUs."""
    import signal
    import time
    tag = f"term{os.getpid()}"
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "HIP_VISIBLE_DEVICES", "LOCAL_WORLD_SIZE",
                                                            "MASTER_ADDR", "MASTER_PORT")}
    env.update(OMP_NUM_THREADS="1", MOY_TEST_TAG=tag)
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "100000", "--warmup", "1", "--dry-run", "--backend", "gloo"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    for _ in range(300):                                           # wait for the two ranks to exist
        kids = [v for v in _pids_with_env(tag) if v != p.pid]
        if len(kids) >= 2:
            break
        time.sleep(0.1)
    assert len(kids) >= 2, kids
    p.send_signal(signal.SIGTERM)
    assert p.wait(timeout=30) == 128 + signal.SIGTERM
    for _ in range(100):
        left = _pids_with_env(tag)
        if not left:
            break
        time.sleep(0.1)
    assert not left, left


def test_synthetic_frames_drawn_with_torch_equal_the_numpy_definition():
    """Round 6: bench.py draws its frames on the device (`SyntheticSequence.frames_torch`: integer torch ops, no host loop -- the
    start-up of a rank no longer costs half a minute of host time); the numpy generator stays the definition and the two are equal
    byte for byte, incl. rectangles that leave the frame and the striping of clipped rectangles."""
    for sid, style, H, W in ((0, "mot17", 608, 1088), (5, "mot17", 96, 160), (1, "dancetrack", 320, 480)):
        sq = SyntheticSequence(sid, H, W, style)
        for t0 in (0, 297, 596):
            n = 4 if H > 400 else 4
            assert np.array_equal(sq.frames(t0, n), sq.frames_torch(t0, n, chunk=3).numpy()), (sid, style, t0)


def test_bench_refuses_a_world_size_that_is_not_gpus():
    """Under a launcher, `--gpus` must equal WORLD_SIZE: one rank reported as eight (or eight as one) is refused with exit code 2."""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=120)
    assert p.returncode == 2 and b"--gpus 2" in p.stderr and not p.stdout.strip()
    p = _bench("--temporal", "100", "--from-host", "--dry-run", timeout=120)     # per-frame-mode legs are rejected with --temporal
    assert p.returncode == 2 and b"not with --temporal" in p.stderr


def test_bench_final_line_is_compact_whatever_the_run_measured():
    """round 4's default line was 23 KB (ten nested legs with full parity blocks) and the driver could not parse it.  The
    final line is built by `compact_line` from the full record: under 4 KB with every block at its largest, json round trip, the
    contract's keys + roofline / cpu_baseline / parity / selfcheck scalars; the rest lives in the side file it names."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    big = "x" * 3000
    stats = {"frames": 32, "topk_overlap": 0.97531, "topk_order_equal_frames": 0, "rows_matched": 9363, "box_max_err_matched": 0.00407823920249939,
             "score_max_err_matched": 0.14741730690002441, "hs_max_err_matched": 0.673367440700531, "births_flipped": 78, "active_rows_reference": 1048,
             "active_rows": 1066, "birth_flip_frac_of_active": 0.07443, "ids_equal": False}
    full = {"metric": "frames/sec (whole node) on 1088x608 MOT17 streams", "value": 15846.81, "unit": "frames/s", "n_gpus": 8, "steps": 20, "warmup": 5,
            "ms_per_step": 36.3480, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": big, "frames_per_step_per_gpu": 576, "streams": 2, "graph": True, "launches_per_step": 83, "weights": big,
                       "sequences_of_rank0": [0], "control_backend": "gloo", "hip_visible_devices_rank0": "0", "dry_run": False, "hbm_allocated_gb": 67.1,
                       "ranks": [{"rank": r, "local_rank": r, "dt_local_s": 0.727123, "fps_local": 15846.81, "hip_visible_devices": str(r), "cpus": "0-31 (32)",
                                  "sequences": [r], "host": big, "device": big, "uuid": big, "pci_bus_id": "0000:05:00.0"} for r in range(8)],
                       "host_link": {"spec": big}, "predictor": big},
            "roofline": {"bound": "hbm", "kernel": "gemm1x1 M2976768 N1536 K128" + big, "launch_index": 40, "achieved": 5417.3, "peak": 8000.0, "unit": "GB/s",
                         "frac": 0.6772, "traffic": 9910000000.0, "traffic_source": big[:200], "avg_ms": 1.8287, "share_of_step": 0.09, "alg_bytes_per_launch": 9907000000,
                         "tflops": 640.0, "runner_up": {"kernel": "stem+conv1 fused" + big, "avg_ms": 1.55, "frac": 0.17, "traffic": 1.0}, "note": big},
            "roofline_step": {"frac": 0.75, "traffic_frac": 0.45, "sum_of_floors_frac": 0.43, "sum_kernel_ms_eager": 19.5, "top_kernels": [{"name": big, "ms": 1.0}] * 8},
            "parity": {"bench_engine_vs_fp32_engine": stats, "token_id_agreement": {"tokens_id_equal_frac": 0.91253, "junk": big},
                       "agreement_hota": {"published": {"HOTA": 37.8, "DetA": 85.0, "AssA": 17.2}, "compat": {"HOTA": 1.0}}, "frames": 32, "first_frame": 8,
                       "bench_engine": "bf16 B=288", "reference_engine": "f32 B=4", "absolute": {"ok": True, "yardstick": {"engine": {"births_flipped": 18, "x": big},
                       "eager_torch_same_dtype": {"births_flipped": 44}}}, "regression": {"ok": True, "bars": {"a": big}}, "ok": True,
                       "fp32_engine_vs_cpu_oracle": {"frames": 2, "logits_max_err": 2.8e-05, "topk_equal": True, "ids_exact": True, "ok": True}},
            "selfcheck": {"ok": True, "replays_bit_identical": True, "value_planes_equal_tiled": True, "detail": {"junk": big}},
            "cpu_baseline": {"value": 10.4, "unit": "frames/s", "cores": 16, "kind": "port", "numeric_only_fps": 10.5, "cores_available": 256, "sample": big},
            "legs": {f"leg{i}": {"value": 1000.0 + i, "parity": big} for i in range(10)}}
    line = json.dumps(bench.compact_line(full, "gpurun_out/bench_full.json"))
    assert len(line) < 4096, len(line)
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d
    assert d["value"] == 15846.81 and d["n_gpus"] == 8 and d["config"]["frames_per_step_per_gpu"] == 576 and len(d["config"]["ranks"]) == 8
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "kernel", "avg_ms", "alg_bytes_per_launch")) <= set(d["roofline"])
    assert d["cpu_baseline"]["value"] == 10.4 and d["cpu_baseline"]["cores"] == 16 and d["cpu_baseline"]["kind"] == "port"
    assert d["parity"]["ok"] is True and d["parity"]["ids_equal"] is False and d["parity"]["birth_flip_frac"] == 0.07443
    assert d["parity"]["births_flipped_engine_vs_eager_same_dtype"] == [18, 44] and d["parity"]["fp32_engine_vs_cpu_oracle"]["ids_exact"] is True
    assert d["selfcheck"] == {"ok": True, "replays_bit_identical": True, "value_planes_equal_tiled": True} and d["full"] == "gpurun_out/bench_full.json"


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



def test_plan_options_are_explicit_and_the_environment_is_a_lab_input_only():
    """Round 6: the plan's A/B switches are fields of PlanOptions -- defaults = the shipped plan, `parse` for
    `bench.py --plan`, `from_env` (the rounds 1-5 variable names) for the lab only; unknown keys are refused."""
    from mo_yolo_amd.engine import PlanOptions
    d = PlanOptions()
    assert d.fold_proj and d.p3_raw and d.dec_tail and d.dec_mid and d.qkv_split == 1 and not d.qkv_fuse and d.qkv_fuse_small
    assert not d.query_order and d.fork_value == 0 and d.fork_small_value == 0 and d.value_planes == 2
    assert d.fold_min_rows == 8192 and d.stem_x3                      # (round 6: the folded head from 13 frames per engine; the f32x3 stem on the matrix cores)
    assert PlanOptions.parse("fold_min_rows=65536,stem_x3=0") == PlanOptions(fold_min_rows=65536, stem_x3=False)
    assert PlanOptions.from_env({"MOY_FOLD_MIN_ROWS": "65536", "MOY_STEM_X3": "0"}) == PlanOptions(fold_min_rows=65536, stem_x3=False)
    p = PlanOptions.parse("p3_raw=0, qkv_split=2,query_order=1,fork_value=128")
    assert (p.p3_raw, p.qkv_split, p.query_order, p.fork_value) == (False, 2, True, 128) and p.fold_proj
    assert PlanOptions.parse("") == d and PlanOptions.parse(None) == d
    with pytest.raises(ValueError):
        PlanOptions.parse("no_such_switch=1")
    e = PlanOptions.from_env({"MOY_FOLD_PROJ": "0", "MOY_QKV_FUSE": "1", "MOY_Q_ORDER": "1", "UNRELATED": "x"})
    assert (e.fold_proj, e.qkv_fuse, e.query_order) == (False, True, True) and e.p3_raw
    assert PlanOptions.from_env({}) == d
    with pytest.raises(Exception):
        d.fold_proj = False                      # frozen: an engine's options cannot change under it

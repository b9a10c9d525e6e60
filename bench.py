#!/usr/bin/env python3
"""Headline benchmark: whole-job frames/s of the per-frame tracking step on synthetic 1088x608 MOT17-shape streams
(BASELINE.json metric; config C2 at N=1, C3 = one sequence shard per GPU at N>1, no collective on the data path).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus 8 --steps 20 --warmup 3            # self-launches 8 ranks (one process per GPU) when not under torchrun
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path (fused preprocess -> backbone/neck -> decoder -> ID assignment + predictor rows) over
one batch of `--batch` frames already resident in HBM (the engines' input slots: a step copies no frame bytes), replayed as
hipGraphs.  Rank 0's LAST stdout line is ONE compact JSON object (< 4 KB: metric, value, roofline, cpu_baseline, parity and
self-check scalars); everything else -- the step roofline three ways, full parity blocks, the launch table -- goes to the side
file the line names (`full`).  Other workloads: `--config c4` (1920x1088, 500 queries), `--config c5` (fp16, 4 sequences batched
per GPU), `--temporal N` (carried track queries, batch element = sequence), `--dtype f16|f32`; `--extra-legs` runs them all as
child processes and prints each as its own short line BEFORE the final one.
`--dry-run --backend gloo` runs the rank -> sequence / barrier / MAX-reduce / rank-0-JSON control flow without a GPU
(CPU test of the N > 1 path).
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic bytes per frame, fused-op convention, bf16 (weights 25.7 MB of it)
ALG_BYTES_FRAME = {"c2": 0.415e9, "c4": 1.154e9}
ALG_WEIGHT_BYTES = 0.0257e9
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable)
MFMA_PEAK_TFLOPS = 2500.0   # dense bf16 / fp16 matrix peak (same guide)
MFMA_F32_PEAK_TFLOPS = 157.3    # fp32-input matrix peak = the fp32 vector peak (same guide): the bound of the exact-fp32 engine
MFMA_SUSTAINED_TFLOPS = 2410.0   # what v_mfma_f32_16x16x32_bf16 sustains on this pool (tools/probes/mfma_peak.hip)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None,
                    help="frames per step per GPU (default 1152 at C2 in 16-bit types, as 4 streams: 288 per sub-batch engine is the largest whose buffers stay "
                         "inside 2 GiB descriptors; 128 at C4; temporal mode: sequences per GPU)")
    ap.add_argument("--config", default="c2", choices=["c2", "c4", "c5", "full"],
                    help="full = yolo_track.yaml at its own depth 1.0 / width 1.0 (the scale the reference's entry script trains, "
                         "start_train.py:11) at 1088x608, 300 queries")
    ap.add_argument("--dtype", default=None, choices=["bf16", "f16", "f32", "f32x3"],
                    help="f32x3 = the fp32 engine with every matrix product in split fp16 precision (MOY_F32X3: ~22 mantissa bits on the 16-bit matrix cores)")
    ap.add_argument("--temporal", type=int, default=0, help="track slots per sequence: carried track queries (DESIGN.md §7)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--streams", type=int, default=None,
                    help="split the batch into this many sub-batches, each on its own HIP stream + hipGraph")
    ap.add_argument("--backend", default="auto", choices=["auto", "gloo", "nccl"],
                    help="torch.distributed backend of the timing barrier / MAX reduce (the data path has no collective)")
    ap.add_argument("--dry-run", action="store_true", help="no GPU: synthetic step time, exercises the N > 1 control flow")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="N > 1 ranks all on GPU 0 of a one-GPU box: runs the real multi-rank code path end to end; its FPS means nothing")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=100, help="CPU-baseline sample: at most this many frames and about 15 s of CPU work")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-launch-table", action="store_true")
    ap.add_argument("--dump-launches", default=None, help="write the per-launch timing table (json) here")
    ap.add_argument("--resize-from", default=None, metavar="HxW",
                    help="frames arrive at this size (e.g. 1080x1920) and every step first stretch-resizes them on the device to the network "
                         "resolution (moy_resize_linear_u8 = LetterBox scaleFill, predict.py:96-105): SURVEY §8(f) rank 3 inside the timed region")
    ap.add_argument("--from-host", action="store_true",
                    help="frames arrive from the HOST inside the timed region: a pinned host ring -> hipMemcpyAsync on a copy stream "
                         "into the next input slot, overlapped with the step (the reference's only device copy, engine/predictor.py:130)")
    ap.add_argument("--predictor", action="store_true",
                    help="time the PRODUCT entry point instead of the engine: TrackPredictor.__call__ on host uint8 frames (pinned ring in, "
                         "one packed device-to-host copy of the rows out, TrackResults built on the host)")
    ap.add_argument("--predictor-calls", action="store_true", help="--predictor: one TrackPredictor.__call__ per step instead of one long stream()")
    ap.add_argument("--pinned-source", action="store_true", help="--predictor: the caller's frames already lie in page-locked memory")
    ap.add_argument("--extra-legs", action="store_true",
                    help="C2 run at N=1: also run the other workloads (C4, C5, temporal, fp32, full scale, host-fed, predictor) as child "
                         "processes, 20 steps each; every leg is printed as its OWN short JSON line before the final line")
    ap.add_argument("--no-extra-legs", action="store_true", help=argparse.SUPPRESS)      # (round 4's spelling of the default)
    ap.add_argument("--fail-rank", type=int, default=None, help=argparse.SUPPRESS)       # test hook: that rank exits 7 before the rendezvous
    ap.add_argument("--host-frames", action="store_true",
                    help="draw the synthetic frames with the numpy generator on the host (the definition) instead of its bit-identical torch form on the device")
    ap.add_argument("--plan", default=None, metavar="KEY=VALUE,...",
                    help="plan options of the engines (mo_yolo_amd.engine.PlanOptions), e.g. query_order=0,p3_raw=0: A/B runs; the default is the shipped plan")
    ap.add_argument("--lab", action="store_true",
                    help="LAB run: load libmoyolo_diag.so (the build that reads the MOY_* A/B knobs and holds the timing-only kernels) and take the plan "
                         "options from the MOY_* environment (PlanOptions.from_env) -- what tools/ab_env.sh and tools/ablation_table.sh pass")
    ap.add_argument("--latency", action="store_true",
                    help="small-batch leg (C5 as BASELINE.json words it: one hipGraph replay = ONE frame of each of --batch live sequences): every step is "
                         "synchronised, the line carries the per-step latency distribution; MOTR/benchmark.py:37-68 times exactly this shape")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="skip the two rocprofv3 --pmc child passes that measure the dominant kernel's HBM traffic in this run (roofline.traffic then "
                         "comes from the committed measurement under profiles/)")
    ap.add_argument("--no-selfcheck", action="store_true", help="skip the determinism / value-planes self-check after the timed region")
    ap.add_argument("--full-out", default=None, help="side file with the full record (default: gpurun_out/bench_full.json if that directory "
                                                      "exists, else bench_full.json beside bench.py)")
    a = ap.parse_args(argv)
    if a.temporal and (a.from_host or a.predictor or a.resize_from):
        ap.error("--from-host / --predictor / --resize-from are per-frame-mode legs: not with --temporal")
    if a.gpus < 1:
        ap.error("--gpus must be >= 1")
    return a


def self_launch(a, argv, timeout_s=None):
    """`bench.py --gpus N` outside torchrun: this process -- which never imports torch and never touches HIP -- starts N ranks of
    itself (one process per GPU, the environment torch.distributed.run would give them), forwards rank 0's stdout, and exits
    non-zero if any rank does.  (ultralytics/utils/dist.py:49-60 builds the same command line for its own multi-GPU entry.)
    Round 6: the ranks are POLLED -- the first non-zero exit (e.g. a rank whose HIP_VISIBLE_DEVICES entry does not exist)
    tears the others down at once instead of leaving rank 0 in the rendezvous until its timeout; SIGTERM / SIGINT to this launcher
    kill the ranks (each in its own process group: its helper threads / children go with it), and the whole job has a deadline."""
    import signal
    import socket
    import subprocess
    import threading
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    n = a.gpus
    deadline = time.monotonic() + (timeout_s if timeout_s is not None else float(os.environ.get("MOY_BENCH_LAUNCH_TIMEOUT_S", "3000")))
    procs = []

    def kill_all(sig=signal.SIGTERM):
        for p in procs:
            if p.poll() is None:
                try:
                    os.killpg(p.pid, sig)              # the rank's own process group (start_new_session below)
                except (ProcessLookupError, PermissionError):
                    pass

    def on_signal(signum, _frame):
        kill_all(signal.SIGTERM)
        time.sleep(0.5)
        kill_all(signal.SIGKILL)
        sys.exit(128 + signum)
    for sg in (signal.SIGTERM, signal.SIGINT):
        signal.signal(sg, on_signal)
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("OMP_NUM_THREADS", "1")
        # every rank pins its own GPU (pin_device: HIP_VISIBLE_DEVICES = its local rank, or its entry of an inherited list) before HIP starts
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env, start_new_session=True,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)     # rank 0's pipe never fills up
    reader.start()
    rcs = [None] * n
    failed = None
    while any(rc is None for rc in rcs):
        for r, p in enumerate(procs):
            if rcs[r] is None:
                rcs[r] = p.poll()
                if rcs[r] not in (None, 0) and failed is None:
                    failed = (r, rcs[r])
        if failed is not None or time.monotonic() > deadline:
            if failed is None:
                failed = ("deadline", 124)
            kill_all(signal.SIGTERM)
            t_end = time.monotonic() + 5.0
            while time.monotonic() < t_end and any(p.poll() is None for p in procs):
                time.sleep(0.05)
            kill_all(signal.SIGKILL)
            rcs = [p.wait() for p in procs]
            break
        time.sleep(0.05)
    reader.join(timeout=5.0)
    sys.stdout.write((out0[0] if out0 else b"").decode())
    sys.stdout.flush()
    if failed is not None:
        print(f"bench.py: rank {failed[0]} failed first (rc {failed[1]}); the other ranks were terminated: rcs = {rcs}", file=sys.stderr, flush=True)
        sys.exit(failed[1] if isinstance(failed[1], int) and failed[1] > 0 else 1)
    sys.exit(0)


def pin_device(local_rank: int, rehearse: bool = False):
    """One process per GPU: make the rank's GPU the only visible one BEFORE anything touches HIP.
    HIP_VISIBLE_DEVICES indexes the devices the ROCr layer exposes (ROCR_VISIBLE_DEVICES narrows that set and renumbers it from
    0), so: a HIP list handed down by the launcher -> this rank takes its local_rank-th entry; otherwise -> the local rank itself,
    whatever ROCR_VISIBLE_DEVICES says."""
    ids = [v for v in os.environ.get("HIP_VISIBLE_DEVICES", "").split(",") if v != ""]
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    if ids:
        # a visibility list handed down by the launcher: this rank takes its local_rank-th entry and NEVER leaves the list
        if local_world > len(ids) and not rehearse:
            raise SystemExit(f"bench.py: {local_world} local ranks but HIP_VISIBLE_DEVICES={','.join(ids)} lists {len(ids)} device(s): "
                             "every rank would land on the same GPU (use --rehearse-one-gpu if that is intended)")
        os.environ["HIP_VISIBLE_DEVICES"] = ids[min(local_rank, len(ids) - 1)]
    else:
        os.environ["HIP_VISIBLE_DEVICES"] = str(local_rank)
    return os.environ["HIP_VISIBLE_DEVICES"]


def pin_cpus(local_rank: int, local_world: int):
    """Host side of one-process-per-GPU: rank r of R local ranks keeps the r-th of R equal slices of the CPUs this job may use
    (in-process `sched_setaffinity`, BEFORE any HIP call or thread pool exists -- never an exec after GPU init), so the launch
    threads of eight ranks do not migrate over each other.  Returns the CPU list as a compact string for the per-rank table."""
    try:
        avail = sorted(os.sched_getaffinity(0))
    except AttributeError:      # pragma: no cover
        return None
    if local_world > 1 and len(avail) >= 2 * local_world and os.environ.get("MOY_BENCH_NO_AFFINITY") != "1":
        per = len(avail) // local_world
        mine = avail[local_rank * per:(local_rank + 1) * per]
        try:
            os.sched_setaffinity(0, mine)
            avail = mine
        except OSError:         # pragma: no cover
            pass
    return f"{avail[0]}-{avail[-1]} ({len(avail)})" if avail else None


def eager_same_dtype_runner(sd, arch):
    """`run(frames_u8, dtype)`: the ORACLE executed in a 16-bit type by eager torch on the frames' device (model and input cast: the
    reference's own `half` switch, engine/predictor.py:131) -- the yardstick of the absolute gate of a 16-bit engine.  Needs only the
    GPU and a few frames, so it runs whether or not the CPU baseline does."""
    import torch
    from oracle import track_oracle as O
    from mo_yolo_amd.synth import to_network_input

    def run16(frames_u8, dt):
        dev = frames_u8.device
        sdh = {k: (v.to(dev, dt) if v.is_floating_point() else v.to(dev)) for k, v in sd.items()}
        with torch.no_grad():
            r = O.forward(to_network_input(frames_u8).to(dt), sdh, arch, anchor_dtype=torch.float32)
        sc = r["dec_scores"].float().sigmoid().max(-1).values
        return dict(topk_ind=r["topk_ind"], boxes=r["dec_bboxes"].float(), scores=sc, obj_idxes=O.assign_ids(sc.cpu()).to(dev),
                    hs=r["hs"].float())
    return run16


def cpu_baseline(cfg, arch, sd, n_frames, engine_check=None):
    """The oracle leg.  (i) Oracle (a port of the reference's eager path, verified against it in the build container) timed on this
    box's host cores, BASELINE.md §4: fp32, warm-ups + a BOUNDED sample (at most n_frames frames and ~15 s of CPU work); FPS for the
    numeric graph only and including the reference-faithful Python state machine.  Round 6: the setting is the best
    the box does among {16, 32, 64, 128} threads at batch 1 and {32, 64, 128} threads on a BATCHED oracle call (8 frames per forward),
    one timed call each; the scan stops climbing when a setting is more than 1.5 x slower than the best so far (all 256 threads:
    124 s per frame, measured in round 3).  (ii) `engine_check(keep)`: the parity gate of BASELINE.md §5 -- the oracle is the checker
    of the fp32 engine on the very frames it is timed on.
    The oracle is the checker / the reported baseline here, never the thing measured as `value`."""
    import torch
    from oracle import track_oracle as O
    from mo_yolo_amd.synth import SyntheticSequence, to_network_input
    seq = SyntheticSequence(0, cfg["H"], cfg["W"], cfg["style"])
    try:
        cores_all = len(os.sched_getaffinity(0))
    except AttributeError:
        cores_all = os.cpu_count() or 1
    t_num = t_state = 0.0
    parity = None
    budget_s = 15.0                                     # bounded sample: the default bench run must finish within a minute
    BATCHED = 8
    with torch.no_grad():
        scan = []
        best = None                                     # (seconds per frame, threads, frames per call)
        for bs in (1, BATCHED):
            x0 = to_network_input(seq.frames(0, bs))
            best_bs = None
            for c in sorted({min(cores_all, v) for v in ((16, 32, 64, 128) if bs == 1 else (32, 64, 128))}):
                torch.set_num_threads(c)
                O.forward(x0, sd, arch)
                t0 = time.perf_counter()
                O.forward(x0, sd, arch)
                spf = (time.perf_counter() - t0) / bs
                scan.append({"threads": c, "frames_per_call": bs, "s_per_frame": round(spf, 4)})
                print(f"[cpu_baseline] {c} threads, {bs} frame(s) per call: {spf:.3f} s/frame", file=sys.stderr, flush=True)
                if best is None or spf < best[0]:
                    best = (spf, c, bs)
                if best_bs is None or spf < best_bs:
                    best_bs = spf
                elif spf > 1.5 * best_bs:
                    break                               # more threads only get slower from here
        _, cores, bs = best
        torch.set_num_threads(cores)
        keep = []
        done = 0
        t_begin = time.perf_counter()
        for i in range(0, n_frames, bs):
            u8 = seq.frames(i, bs)
            x = to_network_input(u8)
            t0 = time.perf_counter()
            r = O.forward(x, sd, arch)
            t1 = time.perf_counter()
            t_num += t1 - t0
            for j in range(bs):
                tj = time.perf_counter()
                scores = r["dec_scores"][j].sigmoid().max(-1).values
                ids, _, _ = O.assign_ids_loop(scores)                       # host loop as shipped (head.py:1232-1243)
                ids = torch.tensor(ids)
                O.tracker_update_copy(scores.tolist(), r["dec_bboxes"][j].numpy(), ids.tolist())
                O.postprocess(r["y"][j], r["dec_scores"][j], ids, 0.25, orig_hw=(cfg["H"], cfg["W"]))
                t_state += time.perf_counter() - tj
                if i + j < 2:
                    keep.append((u8[j:j + 1], {k: v[j:j + 1] for k, v in r.items() if torch.is_tensor(v) and v.shape[:1] == (bs,)}, ids))
            done += bs
            if done % 24 == 0:
                print(f"[cpu_baseline] {done}/{n_frames} frames", file=sys.stderr, flush=True)
            if time.perf_counter() - t_begin > budget_s and done >= 5:
                break
        n_frames = done
        if engine_check is not None:
            parity = engine_check(keep)
    out = {"value": round(n_frames / (t_num + t_state), 3), "unit": "frames/s", "cores": cores, "kind": "port",
           "numeric_only_fps": round(n_frames / t_num, 3),
           "cores_available": cores_all, "frames_per_call": bs, "scan": scan,
           "sample": f"{n_frames} frames ({t_num + t_state:.1f} s of CPU work; best of a scan over 16-128 threads x 1 | {BATCHED} frames per call: "
                     f"{cores} threads, {bs} per call) of the same stream, fp32 eager torch-CPU oracle incl. the host state machine"}
    return out, parity


def live_traffic(kernel_substr, bench_args, out_dir, budget_s=170.0):
    """HBM bytes per launch of one kernel from PMC counters collected IN THIS RUN (`roofline.traffic` used to be a committed
    constant): two child runs of this script under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (SEPARATE passes, and
    --kernel-trace the only trace domain beside them: MI355X_MICROARCH.md, HBM section) on one sub-batch engine, one stream, eager launches;
    bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 -- the guide's gfx950 correction (FETCH_SIZE counts 64 bytes per 128-byte request of a wide
    read; tools/traffic_summary.py).  The children are fresh processes (rocprofv3 -> python3 bench.py: no exec after GPU init).  Returns
    (bytes or None, source text); any failure or time-out returns None and the caller keeps the committed constant."""
    import csv
    import glob
    import shutil
    import subprocess
    rp = shutil.which("rocprofv3")
    if rp is None:
        return None, "rocprofv3 not on PATH"
    t_end = time.monotonic() + budget_s
    vals = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(out_dir, f"live_pmc_{ctr.lower()}")
        shutil.rmtree(d, ignore_errors=True)
        left = t_end - time.monotonic()
        if left < 20:
            return None, "time budget of the live PMC passes used up"
        cmd = [rp, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__), *bench_args]
        try:
            env = dict(os.environ, TMPDIR="/tmp")
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=left)
        except subprocess.TimeoutExpired:
            return None, f"the {ctr} pass timed out"
        if r.returncode != 0:
            return None, f"the {ctr} pass failed (rc {r.returncode})"
        got = []
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if row.get("Counter_Name") == ctr and kernel_substr in row.get("Kernel_Name", ""):
                    got.append(float(row["Counter_Value"]))
        shutil.rmtree(d, ignore_errors=True)
        if not got:
            return None, f"no {ctr} rows for {kernel_substr}"
        vals[ctr] = (sum(got) / len(got), len(got))
    by = (2.0 * vals["FETCH_SIZE"][0] + vals["WRITE_SIZE"][0]) * 1024.0
    return by, (f"PMC counters of THIS run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate child passes, {kernel_substr}, {vals['FETCH_SIZE'][1]} dispatches, "
                f"(2*FETCH_SIZE + WRITE_SIZE)*1024, fetch {vals['FETCH_SIZE'][0]:.0f} KB raw, write {vals['WRITE_SIZE'][0]:.0f} KB")


def copy_peak_gbs(dev):
    """Box-measured copy peak: device-to-device copy of 1 GiB, read + write bytes per second."""
    import torch
    n = 1 << 30
    a = torch.empty(n, dtype=torch.uint8, device=dev)
    b = torch.empty(n, dtype=torch.uint8, device=dev)
    b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    del a, b
    return 2 * n / (ms * 1e-3) / 1e9


def h2d_peak_gbs(dev):
    """Box-measured host-to-device rate: 1 GiB from pinned host memory, hipMemcpyAsync on the current stream."""
    import torch
    n = 1 << 30
    h = torch.empty(n, dtype=torch.uint8).pin_memory()
    d = torch.empty(n, dtype=torch.uint8, device=dev)
    d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        d.copy_(h, non_blocking=True)
    e1.record()
    torch.cuda.synchronize()
    return n / (e0.elapsed_time(e1) / 3 * 1e-3) / 1e9


def log(msg):
    """Progress on stderr (a long silent run is taken to be hung by the GPU pool's watchdog)."""
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main(argv=None):
    t_process = time.perf_counter()
    a = parse(argv)
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        self_launch(a, list(sys.argv[1:] if argv is None else argv))          # never returns
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != a.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {a.gpus} but the launcher started {world} rank(s) (WORLD_SIZE): refusing to report one as the other",
                  file=sys.stderr, flush=True)
        sys.exit(2)
    local = int(os.environ.get("LOCAL_RANK", 0))
    if a.fail_rank is not None and rank == a.fail_rank:
        sys.exit(7)
    visible = None
    # (the dry run pins too -- it only writes the environment variable -- so the exact rank -> device map is observable without a GPU)
    visible = pin_device(0 if a.rehearse_one_gpu else local, a.rehearse_one_gpu)
    cpus = pin_cpus(local, int(os.environ.get("LOCAL_WORLD_SIZE", world)))      # before anything touches HIP / starts threads

    if a.lab and not a.dry_run:                          # before anything imports mo_yolo_amd._lib
        os.environ.setdefault("MOYOLO_LIB", os.path.join(ROOT, "mo_yolo_amd", "libmoyolo_diag.so"))
    import torch
    from mo_yolo_amd import shard
    dist = None
    backend = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # Control plane only (timing barrier, MAX over ranks): gloo over loopback needs no peer access between the pinned
        # devices; `--backend nccl` (= RCCL) is available for a node where it is preferred.
        backend = "gloo" if a.backend in ("auto", "gloo") else "nccl"
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
        else:
            dist.init_process_group("gloo")

    # ---- workload
    cfg_name = {"c5": "c2", "full": "full_c2"}.get(a.config, a.config)
    dtype_name = a.dtype or ("f16" if a.config == "c5" else "bf16")
    seq_per_gpu = 4 if (a.config == "c5" or a.temporal) else 1
    if a.temporal:
        B = a.batch or 4
        seq_per_gpu = B
        S = 1
    elif a.latency:
        # the small-batch reading of C5: one replay = ONE frame of each of B live sequences
        B = a.batch or 4
        seq_per_gpu = B
        S = 1
    else:
        # (fp32 buffers are twice the size: 96 frames per engine; the full-width model's widest buffer -- the first C2f's
        #  [y0 | y1 | y2 | y3 | y4] at 152 x 272 x 320 channels -- allows 81 per engine inside 2 GiB descriptors; 76 makes the 256-row tiles of
        #  the 38 x 68 level fill exactly three rounds of 256 CUs: +1.3 % over 64)
        B = a.batch or {"c4": 128, "full_c2": 152}.get(cfg_name, 192 if dtype_name in ("f32", "f32x3") else 576)
        if a.predictor:
            B = a.batch or 288
        S = max(1, a.streams if a.streams is not None else 2)
        # Round 5: the 16-bit C2 / C5 default is FOUR sub-batch engines of 288 frames on four streams (88.6 GB of the 288 GB): same-device
        # 17.54 k (2 x 288) -> 17.69 k (3) -> 17.93 k (4) -> 17.81 k (6) frames/s, tools/ab_batch.sh.  An explicit --batch or --streams keeps its meaning.
        # C4 (64 frames per engine): 6.09 k (2) -> 6.19 k (3) -> 6.26 k (4); the 1.0 / 1.0 scale: 3.71 -> 3.73 k, left at two.
        if (a.batch is None and a.streams is None and cfg_name in ("c2", "c4") and dtype_name in ("bf16", "f16") and not a.predictor
                and not a.from_host and not a.resize_from):
            B, S = (1152, 4) if cfg_name == "c2" else (256, 4)
        # Round 6: every benched frame lies INSIDE its 600-frame sequence -- a batch of more than 576 frames is cut from
        # ceil(B / 576) sequences per GPU (the default 1152: frames 0..575 of two sequences, sub-batch engines 0, 1 on the first, 2, 3 on
        # the second; C3 at N GPUs then runs 2 N sequences, two per GPU: still no cross-GPU term)
        if a.config != "c5" and not a.predictor:
            seq_per_gpu = max(1, -(-B // 576))
    if B % S or (not a.temporal and B % seq_per_gpu):
        raise SystemExit("--batch must be a multiple of --streams and of the sequences per GPU")
    frames_step = B                      # frames one timed step processes on this rank (the predictor leg: several chunks of B)
    # sequence shard of this rank (sequence i -> rank i mod N, no cross-GPU term)
    my_seqs = shard.sequences_for_rank(world * seq_per_gpu, rank, world)

    def barrier():
        if not a.dry_run:
            torch.cuda.synchronize()           # first: no rank leaves the barrier with work in flight
        if world > 1:
            dist.barrier()

    line_extra = {}
    if a.dry_run and world > 1:
        got = [None] * world
        dist.all_gather_object(got, {"rank": rank, "local_rank": local, "hip_visible_devices": visible, "sequences": my_seqs})
        line_extra["rank_map"] = got
    if a.dry_run:
        def step(i):
            time.sleep(0.002 * (1 + 0.1 * rank))       # synthetic, rank dependent: the MAX over ranks is observable
        eng = None
    else:
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        from mo_yolo_amd.engine import StreamedEngines, TrackEngine
        from mo_yolo_amd.fixtures import fixture
        from mo_yolo_amd.synth import SyntheticSequence
        cfg, arch, sd = fixture(cfg_name)
        dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32, "f32x3": torch.float32}[dtype_name]
        ekw = dict(split_f16=True) if dtype_name == "f32x3" else {}
        if a.plan or a.lab:
            from mo_yolo_amd.engine import PlanOptions
            ekw["options"] = PlanOptions.parse(a.plan) if a.plan else PlanOptions.from_env()
            line_extra["plan_options"] = a.plan or ("from the MOY_* environment: " + ",".join(f"{k}={os.environ[v]}" for k, v in PlanOptions._ENV.items() if v in os.environ))
        if a.lab:
            line_extra["library"] = os.environ.get("MOYOLO_LIB")
        seqs = [SyntheticSequence(sid, cfg["H"], cfg["W"], cfg["style"]) for sid in my_seqs]
        n_slots = 3

        def seq_frames(sq, t0, n):
            """frames t0 .. t0+n-1 of a sequence as a uint8 device tensor: drawn on the device (bit-identical torch form of the numpy
            generator: no host loop, start-up does not scale with the ranks of a node) unless --host-frames"""
            if a.host_frames:
                return torch.from_numpy(sq.frames(t0, n)).to(dev)
            return sq.frames_torch(t0, n, device=dev)

        def batch_frames(i):
            """Step i's frames: per-frame mode -> B/seq_per_gpu consecutive frames of every sequence of this rank;
            temporal / latency mode -> frame i of each of the B sequences."""
            if a.temporal or a.latency:
                return torch.cat([seq_frames(s, i, 1) for s in seqs])
            per = B // len(seqs)
            # the synthetic sequences are 600 frames long by design: slot i starts at frame i * per while that stays
            # inside the sequence, else the slots are windows 8 frames apart -- per <= 576, so every window ends before frame 600
            t0 = i * per if (i + 1) * per <= 600 else 8 * i
            assert t0 + per <= 600, "a benched frame outside its 600-frame sequence"
            return torch.cat([seq_frames(s, t0, per) for s in seqs])

        if a.temporal:
            eng = TrackEngine(arch, sd, cfg["H"], cfg["W"], batch=B, dtype=dtype, device=dev, temporal=a.temporal, n_inputs=n_slots, **ekw)
            for k in range(n_slots):
                eng.inputs[k].copy_(batch_frames(k))
            eng.forward(slot=0)
            torch.cuda.synchronize()
            if not a.no_graph:
                eng.capture()
            eng.reset_sequence()

            def step(i):
                eng.forward(slot=i % n_slots)
            pipe = None
        elif a.predictor:
            # the product entry point (mo_yolo_amd/predictor.py): host uint8 frames in, TrackResults out
            from mo_yolo_amd.predictor import TrackPredictor
            pipe = None
            pred = TrackPredictor(arch, sd, imgsz=(cfg["H"], cfg["W"]), dtype=dtype, device=dev, batch=B, graph=not a.no_graph,
                                  streams=a.streams if a.streams is not None else 2)
            n_chunks = 6
            host_frames = torch.cat([batch_frames(k % 2).cpu() for k in range(n_chunks)])
            if a.pinned_source:
                host_frames = host_frames.pin_memory()               # a decoder that writes into page-locked memory: no staging copy
            else:
                host_frames = host_frames.numpy()
            pred(host_frames[:B])                                # builds the engine, the ring and the graphs
            eng = next(iter(pred._engines.values()))
            line_extra["predictor"] = ((f"TrackPredictor.__call__ on {n_chunks * B} host frames per step" if a.predictor_calls else
                                        f"TrackPredictor.stream() over arrays of {n_chunks * B} host frames (one array per step)")
                                       + f" (chunks of {B}), source = "
                                       + ("one PINNED uint8 tensor (no staging copy)" if a.pinned_source else
                                          "a pageable numpy array (staged into the pinned ring by 8 host threads)")
                                       + " -> H2D on a copy stream -> step -> ONE packed D2H -> TrackResults on the host")
            n_results = [0]
            if a.predictor_calls:
                def step(i):
                    n_results[0] = len(pred(host_frames))           # one __call__ per step: the pipeline fills and drains every time
            else:
                def forever():
                    while True:
                        yield host_frames
                results_gen = pred.stream(forever())                # the generator form: the pipeline stays full across the steps

                def step(i):
                    for _ in range(n_chunks):
                        n_results[0] += len(next(results_gen))
            frames_step = n_chunks * B                           # frames per timed step (B stays the chunk = the engines' batch)
        else:
            pipe = StreamedEngines(arch, sd, cfg["H"], cfg["W"], batch=B, streams=S, graph=not a.no_graph, dtype=dtype, device=dev,
                                   n_inputs=n_slots, **ekw)
            eng = pipe.engines[0]
            for k in range(n_slots):
                pipe.load(batch_frames(k), slot=k)       # frames resident in HBM before the timed region: no copy inside a step
            torch.cuda.synchronize()
            pipe.forward(slot=0)                          # first call: eager pass + graph capture
            torch.cuda.synchronize()
            if a.resize_from and not a.from_host:
                # camera-resolution frames resident in HBM (one source slot per engine: the same pixels every step; only their
                # SIZE matters to the resize kernel), resized on each engine's stream into the input slot its graph then reads
                from mo_yolo_amd import ops as _ops
                hs, ws = (int(v) for v in a.resize_from.lower().split("x"))
                gsrc = torch.Generator(device="cpu").manual_seed(7)
                src = [torch.randint(0, 256, (pipe.Bs, hs, ws, 3), dtype=torch.uint8, generator=gsrc).to(dev) for _ in pipe.engines]
                line_extra["resize_from"] = f"{ws}x{hs}"

                def step(i):
                    slot = i % n_slots
                    cur_s = torch.cuda.current_stream()
                    for e, st_, sr in zip(pipe.engines, pipe.streams, src):
                        st_.wait_stream(cur_s)
                        with torch.cuda.stream(st_):
                            _ops.resize_linear_u8(sr, (cfg["H"], cfg["W"]), out=e.inputs[slot])
                    pipe.forward(slot=slot)
            elif a.from_host:
                # the frames of step i+1 cross PCIe while step i computes.  Pinned host ring (one buffer per input
                # slot and engine, filled before the timed region: the decoder's side of the ring) -> hipMemcpyAsync on a copy
                # stream into input slot (i+1) % 3 -> that engine's stream waits for the copy's event.  With --resize-from the ring
                # holds camera-resolution frames, which land in device staging buffers and are stretch-resized into the slot.
                from mo_yolo_amd import ops as _ops
                hs, ws = (int(v) for v in a.resize_from.lower().split("x")) if a.resize_from else (cfg["H"], cfg["W"])
                line_extra["from_host"] = f"pinned ring, {ws}x{hs} uint8 BGR frames, {hs * ws * 3 / 1e6:.2f} MB per frame"
                g7 = torch.Generator(device="cpu").manual_seed(7)
                if a.resize_from:    # camera-resolution frames: ONE pinned buffer per engine serves its three slots (only the bytes' SIZE
                    # matters to the copy and to the resize kernel; 10.7 GB of random bytes would take the leg half a minute to draw)
                    one = [torch.randint(0, 256, (pipe.Bs, hs, ws, 3), dtype=torch.uint8, generator=g7).pin_memory() for _ in pipe.engines]
                    host = [[one[i]] * n_slots for i in range(len(pipe.engines))]
                else:
                    host = [[e.inputs[k].cpu().pin_memory() for k in range(n_slots)] for e in pipe.engines]
                stage = [[torch.empty(pipe.Bs, hs, ws, 3, dtype=torch.uint8, device=dev) for _ in range(n_slots)] if a.resize_from
                         else e.inputs for e in pipe.engines]
                copy_streams = [torch.cuda.Stream(device=dev) for _ in pipe.engines]
                copied = [[None] * n_slots for _ in pipe.engines]      # event: the slot's frames have arrived
                consumed = [[None] * n_slots for _ in pipe.engines]    # event: the step that read the slot has finished

                copy_timing = []                                        # (start, end) timed events around every copy: the link rate as observed

                def enqueue_copy(k, slot):
                    cs = copy_streams[k]
                    with torch.cuda.stream(cs):
                        if consumed[k][slot] is not None:
                            cs.wait_event(consumed[k][slot])
                        e0 = torch.cuda.Event(enable_timing=True)
                        e0.record(cs)
                        stage[k][slot].copy_(host[k][slot], non_blocking=True)
                        ev = torch.cuda.Event(enable_timing=True)
                        ev.record(cs)
                        copied[k][slot] = ev
                        copy_timing.append((e0, ev))

                for k in range(len(pipe.engines)):
                    enqueue_copy(k, 0)

                def step(i):
                    slot, nxt = i % n_slots, (i + 1) % n_slots
                    for k, (e, st_) in enumerate(zip(pipe.engines, pipe.streams)):
                        st_.wait_event(copied[k][slot])
                        if a.resize_from:
                            with torch.cuda.stream(st_):
                                _ops.resize_linear_u8(stage[k][slot], (cfg["H"], cfg["W"]), out=e.inputs[slot])
                    pipe.forward(slot=slot)
                    for k, st_ in enumerate(pipe.streams):
                        ev = torch.cuda.Event()
                        ev.record(st_)
                        consumed[k][slot] = ev
                        enqueue_copy(k, nxt)                       # next step's frames cross the link beside this step
            else:
                def step(i):
                    pipe.forward(slot=i % n_slots)

    if rank == 0:
        log("plan built and captured; warm-up")
    startup_s = time.perf_counter() - t_process         # process start -> plan built, frames resident, graphs captured (per rank)
    for i in range(a.warmup):
        step(i)
    barrier()
    lat = []
    t0 = time.perf_counter()
    if a.latency and not a.dry_run:
        # small-batch leg: the caller needs frame t's tracks before frame t+1 exists, so every step is followed by a device
        # synchronise (the shape MOTR/benchmark.py:37-68 and ops.Profile time); still EXACTLY K steps between the two barriers
        for i in range(a.steps):
            ts = time.perf_counter()
            step(i)
            torch.cuda.synchronize()
            lat.append(time.perf_counter() - ts)
    else:
        for i in range(a.steps):
            step(i)
    barrier()
    dt_local = time.perf_counter() - t0
    if lat:
        ls = sorted(lat)
        line_extra["latency_ms"] = {"mean": round(sum(ls) / len(ls) * 1e3, 4), "p50": round(ls[len(ls) // 2] * 1e3, 4),
                                    "p99": round(ls[min(len(ls) - 1, int(len(ls) * 0.99))] * 1e3, 4), "min": round(ls[0] * 1e3, 4),
                                    "frames_per_step": frames_step, "synchronised_every_step": True}
    try:
        import resource
        rss_mb = round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0, 1)
    except Exception:  # pragma: no cover
        rss_mb = None
    line_extra["startup_s_rank0"] = round(startup_s, 2)
    line_extra["host_rss_mb_rank0"] = rss_mb
    dt = shard.max_over_ranks(dt_local, device=(torch.device("cuda", 0) if backend == "nccl" else None))
    fps = shard.whole_job_fps(frames_step * a.steps, dt, world)
    if world > 1:
        # the first hardware multi-GPU run must be diagnosable -- every rank's own time and device, not only the MAX
        me = {"rank": rank, "local_rank": local, "dt_local_s": round(dt_local, 6), "fps_local": round(frames_step * a.steps / dt_local, 2),
              "hip_visible_devices": visible, "cpus": cpus, "sequences": my_seqs, "host": os.uname().nodename,
              "startup_s": round(startup_s, 2), "host_rss_mb": rss_mb}
        if not a.dry_run:
            pr = torch.cuda.get_device_properties(0)
            me.update(device=pr.name, gcn_arch=getattr(pr, "gcnArchName", None), cus=pr.multi_processor_count,
                      uuid=str(getattr(pr, "uuid", "")), pci_bus_id=getattr(pr, "pci_bus_id", None))
        line_extra["ranks"] = sorted(shard.gather_objects(me), key=lambda e: e["rank"])
    if rank == 0:
        log(f"timed region done: {fps:.1f} frames/s")

    roof = roof_step = parity = None
    cpu = None
    if rank == 0 and not a.dry_run and (a.from_host or a.predictor):
        # what crossed the host link inside the timed region, next to what the link of THIS box moves when it does nothing else
        fb = (hs * ws * 3) if a.from_host else cfg["H"] * cfg["W"] * 3
        try:
            link = round(h2d_peak_gbs(dev), 1)
        except Exception as e:  # pragma: no cover
            link = repr(e)
        if a.from_host:
            cms = [e0.elapsed_time(e1) for e0, e1 in copy_timing[-2 * a.steps:]]
            line_extra["host_copies"] = {"copies_timed": len(cms), "bytes_each": int(host[0][0].numel()),
                                         "ms_each_mean": round(sum(cms) / len(cms), 3), "ms_each_min": round(min(cms), 3),
                                         "gbs_each_mean": round(host[0][0].numel() / (sum(cms) / len(cms) * 1e-3) / 1e9, 2)}
        line_extra["host_link"] = {"h2d_gbs_in_timed_region": round(fps / world * fb / 1e9, 2), "h2d_peak_gbs_measured": link,
                                   "bytes_per_frame": fb, "spec": "PCIe Gen5 x16, 63 GB/s per direction (MI355X_MICROARCH.md)"}
    if rank == 0 and not a.dry_run:
        out = eng.outputs()
        Bs = eng.B
        n_masked = int(out["n_masked"].sum())
        active = float((out["obj_idxes"] >= 0).sum()) / Bs
        line_extra.update(mean_active_tracks=round(active, 1), masked_tokens_selected=n_masked,
                          hbm_allocated_gb=round(torch.cuda.max_memory_allocated() / 1e9, 1))

        # ---- per-launch timing with HIP events on the launch stream (eager replay of the same plan, one sub-batch engine)
        if not a.no_launch_table:
            st = torch.cuda.current_stream()
            nL = eng.num_launches
            acc = [0.0] * nL
            reps = max(3, min(a.steps, 10))
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(nL + 1)]
            for rep in range(reps):
                evs[0].record(st)
                for i in range(nL):
                    eng.run_steps(i, i + 1, slot=rep % n_slots)
                    evs[i + 1].record(st)
                torch.cuda.synchronize()
                for i in range(nL):
                    acc[i] += evs[i].elapsed_time(evs[i + 1])
            per = [x / reps for x in acc]                              # ms per launch
            prof = {}
            tp = os.path.join(ROOT, "profiles", "traffic_by_launch.json")
            if os.path.exists(tp):
                prof = json.load(open(tp))
            plaunch = prof.get("launches", prof)

            # The DOMINANT kernel as `rocprofv3 --stats` ranks kernels: by TOTAL time over its calls in a pass, not by its longest
            # single launch (round 5: the deformable gather runs six times per pass).  Launches with the same name are the same kernel
            # on the same shape; `avg_ms` is the mean over those calls -- the figure the committed kernel-stats csv shows for it.
            groups = {}
            for i in range(nL):
                groups.setdefault(eng.meta[i]["name"], []).append(i)

            def kernel_record(name):
                idx = groups[name]
                avg = sum(per[i] for i in idx) / len(idx)
                mm = eng.meta[idx[0]]
                by, fl = mm["bytes"], mm["flops"]
                mpk = MFMA_F32_PEAK_TFLOPS if dtype_name == "f32" else (MFMA_PEAK_TFLOPS / 3 if dtype_name == "f32x3" else MFMA_PEAK_TFLOPS)
                gbs, tf = (by / (avg * 1e-3) / 1e9 if by else 0.0), fl / (avg * 1e-3) / 1e12
                rec = {"kernel": name, "calls_per_pass": len(idx), "avg_ms": round(avg, 4), "total_ms_per_pass": round(avg * len(idx), 4),
                       "share_of_pass": round(avg * len(idx) / sum(per), 4), "alg_bytes_per_launch": by, "alg_flops_per_launch": fl,
                       "hbm_gbs": round(gbs, 1), "tflops": round(tf, 2)}
                # the roof that bounds it: whichever floor (bytes / HBM peak, flops / matrix peak of the arithmetic type) is the longer
                if fl / (mpk * 1e12) > by / (HBM_PEAK_GBS * 1e9):
                    rec.update(bound="mfma", achieved=round(tf, 2), peak=round(mpk, 1), unit="TFLOP/s", frac=round(tf / mpk, 4))
                else:
                    rec.update(bound="hbm", achieved=round(gbs, 1), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(gbs / HBM_PEAK_GBS, 4))
                tr = plaunch.get(name, {}).get("hbm_bytes") if dtype_name == "bf16" else None      # (measured on the bf16 plan)
                rec["traffic"] = tr
                # Round 6 (VERDICT r5 #2 / ADVICE r5): next to `frac` (ALGORITHMIC bytes or flops over the launch time: the contract's
                # figure) the same launch time against the bytes the PMC passes saw cross HBM -- for the gather the algorithmic figure is
                # an expectation model (distinct cells under uniform sampling) and the queries of a frame cluster, so it overstates
                rec["alg_frac"] = rec["frac"]
                rec["traffic_frac"] = round(tr / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if tr else None
                # (a committed constant: PMC passes of an earlier run of this plan, NOT a counter of this run)
                rec["traffic_source"] = ("committed PMC measurement, not a counter of this run: "
                                         + str(plaunch.get(name, {}).get("source", "profiles/traffic_by_launch.json"))[:160]) if tr is not None else None
                return rec
            ranked = sorted(groups, key=lambda n: -sum(per[i] for i in groups[n]))
            roof = kernel_record(ranked[0])
            roof["dominant_by"] = "total time over the kernel's calls in a pass (rocprofv3 --stats order)"
            if len(ranked) > 1:
                ru = kernel_record(ranked[1])
                roof["runner_up"] = {k: ru[k] for k in ("kernel", "calls_per_pass", "avg_ms", "total_ms_per_pass", "bound", "achieved", "frac", "traffic", "traffic_frac")}
            longest = max(range(nL), key=lambda i: per[i])
            roof["longest_single_launch"] = {"kernel": eng.meta[longest]["name"], "ms": round(per[longest], 4)}
            # `traffic` of the dominant kernel from counters of THIS run (two child passes, ~1 min; the committed constant stays as the fallback)
            sym = next((v for k, v in (("msda_raw0", "msda_raw_mfma_kernel"), ("stem+conv1", "stem_l1_kernel"), ("c2f fused", "c2f_fused_kernel"))
                        if ranked[0].startswith(k)), None)
            under_profiler = any("rocprof" in os.environ.get(k, "").lower() for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB"))
            if (sym and world == 1 and not a.no_live_traffic and not under_profiler and not a.temporal and not a.predictor and not a.from_host and not a.resize_from
                    and not a.latency and dtype_name in ("bf16", "f16") and a.config in ("c2", "c5")):
                log(f"live PMC passes: HBM traffic of {sym} (two rocprofv3 child runs)")
                child = ["--batch", str(eng.B), "--streams", "1", "--no-graph", "--no-cpu-baseline", "--no-parity", "--no-launch-table", "--no-selfcheck",
                         "--no-live-traffic", "--steps", "2", "--warmup", "1", "--dtype", dtype_name, "--full-out", os.path.join(os.path.dirname(full_path(a)), "bench_full_live_pmc_child.json")]
                if a.plan:
                    child += ["--plan", a.plan]
                by_live, src_live = live_traffic(sym, child, os.path.dirname(full_path(a)))
                if by_live:
                    roof["traffic"] = by_live
                    roof["traffic_frac"] = round(by_live / (roof["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                    roof["traffic_source"] = src_live
                else:
                    roof["traffic_live_error"] = src_live
            if ranked[0].startswith("stem"):
                roof["note"] = ("fused preprocess + stem + conv1: bound by vector-instruction issue (SiLU at 28 cycles per value, uint8 fragment build), "
                                "neither the HBM nor the matrix roof is near")
            if ranked[0].startswith("msda_raw0"):
                roof["limiter"] = "l2->l1 gather rate / vector-memory issue (NOT HBM: see traffic_frac)"
                roof["note"] = ("deformable gather, level 0 raw: `bound` names the larger of its two floors (bytes), but what limits it is the rate at which the "
                                "memory system serves its 2 x 2-window taps from L2 into L1 (about 5 x its HBM bytes cross L2 -> L1), DESIGN.md section 4")
            ms_step = dt / a.steps * 1e3
            n_eng = 1 if pipe is None else len(pipe.engines)
            plan_bytes = sum(mm["bytes"] for mm in eng.meta) * frames_step // eng.B      # (every engine's plan is the same; a step = frames_step / eng.B passes)
            roof_step = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "launches": nL, "sum_kernel_ms_eager": round(sum(per), 3),
                         # (b) what the plan's launches move by their own fused-op accounting (every launch kind has a byte count)
                         "plan_bytes_per_step": plan_bytes,
                         "plan_frac": round(plan_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            if cfg_name in ALG_BYTES_FRAME and not a.temporal:
                # (a) SURVEY §8(d) convention: layer-wise algorithmic bytes of the reference's op list (the figure the 60 % target
                # is stated in; it charges value_proj's input six times and enc_output over all S tokens, which this plan avoids)
                bytes_step = (ALG_BYTES_FRAME[cfg_name] - ALG_WEIGHT_BYTES) * frames_step + ALG_WEIGHT_BYTES * max(1, frames_step // eng.B)
                if dtype_name in ("f32", "f32x3"):
                    bytes_step *= 2
                ach_s = bytes_step / (ms_step * 1e-3) / 1e9
                roof_step.update(achieved=round(ach_s, 1), frac=round(ach_s / HBM_PEAK_GBS, 4), alg_bytes_per_step=bytes_step)
                # SURVEY §8(d) as written charges the 25.7 MB of weights to EVERY frame (FPS x B_alg / peak: 11.6 k FPS = 0.60 at C2);
                # `frac` above charges them once per launch (they are read once per sub-batch), the stricter figure
                roof_step["frac_survey_8d_per_frame_weights"] = round(ALG_BYTES_FRAME[cfg_name] * (2 if dtype_name in ("f32", "f32x3") else 1) * frames_step
                                                                      / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            # (c) measured HBM traffic of the plan (PMC passes committed under profiles/, tools/pmc_traffic.sh): bytes per frame
            tpf = prof.get("step_total", {}).get(f"{cfg_name}_{dtype_name}", {}).get("hbm_bytes_per_frame")
            if tpf and not a.temporal:                           # (measured on the per-frame plan)
                tb = tpf * frames_step
                roof_step.update(traffic_bytes_per_step=tb, traffic_frac=round(tb / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                 traffic_source=prof["step_total"][f"{cfg_name}_{dtype_name}"].get("source"))
            try:
                roof_step["copy_peak_gbs_measured"] = round(copy_peak_gbs(dev), 1)
            except Exception as e:  # pragma: no cover
                roof_step["copy_peak_gbs_measured"] = repr(e)
            if a.dump_launches:
                with open(a.dump_launches, "w") as f:
                    json.dump([dict(i=i, ms=per[i], **eng.meta[i]) for i in range(nL)], f)
            # (d) sum over the launches of max(bytes / HBM peak, flops / dense MFMA peak) against the time they take one after the
            # other: the fraction of the step's kernel time that the launches' OWN floors account for (0.43)
            # (f32x3: three fp16 products per fp32-equivalent product)
            mfma_peak = MFMA_F32_PEAK_TFLOPS if dtype_name == "f32" else (MFMA_PEAK_TFLOPS / 3 if dtype_name == "f32x3" else MFMA_PEAK_TFLOPS)
            floors = [max(mm["bytes"] / (HBM_PEAK_GBS * 1e9), mm["flops"] / (mfma_peak * 1e12)) * 1e3 for mm in eng.meta]
            roof_step["sum_of_launch_floors_ms"] = round(sum(floors), 3)
            roof_step["sum_of_floors_frac"] = round(sum(floors) / max(sum(per), 1e-9), 4)
            # the same against what THIS device sustains: the copy rate measured above (a stream that reads and writes does not get the
            # fill rate: DESIGN.md round-3 item 12) and the sustained MFMA rate -- informative, never `roofline.frac`
            cp = roof_step.get("copy_peak_gbs_measured")
            if isinstance(cp, float) and cp > 0:
                msus = 155.0 if dtype_name == "f32" else (MFMA_SUSTAINED_TFLOPS / 3 if dtype_name == "f32x3" else MFMA_SUSTAINED_TFLOPS)
                sf = [max(mm["bytes"] / (cp * 1e9), mm["flops"] / (msus * 1e12)) * 1e3 for mm in eng.meta]
                roof_step["sum_of_sustained_floors_frac"] = round(sum(sf) / max(sum(per), 1e-9), 4)
            worst_l = max(range(nL), key=lambda i: floors[i] / max(per[i], 1e-9))
            roof_step["max_launch_floor_frac"] = {"name": eng.meta[worst_l]["name"], "frac": round(floors[worst_l] / max(per[worst_l], 1e-9), 4)}
            top = sorted(range(nL), key=lambda i: -per[i])[:8]
            roof_step["top_kernels"] = [{"name": eng.meta[i]["name"], "ms": round(per[i], 4)} for i in top]

        # ---- parity gates recorded with the run: the engine AS BENCHED (its batch, dtype, kernels
        # selected at that launch size) against a small-batch fp32 engine of the same weights on the same frames
        if not a.no_parity and not a.temporal:
            log("parity gate: benched engine vs small-batch fp32 engine")
            from mo_yolo_amd.parity import agreement_hota, engine_pair_stats, token_id_agreement, tracks_of
            # 32 frames of the stream AFTER the fixture frames (on frames 0..7 the calibration parks the rows it moved exactly at the
            # edge of the threshold bands, one sigma of the bf16 logit noise away: they flip twice as often as stream rows do)
            NP, NB, P0 = min(32, Bs - 8), 4, 8
            small = NP < NB                    # a small-batch engine (--latency, --batch 4): the window goes through it in chunks of its batch
            if small:
                NP = 32 - 32 % (NB * Bs // math.gcd(NB, Bs))
            NP -= NP % NB
            ref = TrackEngine(arch, sd, cfg["H"], cfg["W"], batch=NB, dtype=torch.float32, device=dev)
            if small:
                fr = seq_frames(seqs[0], P0, NP)
                chunks = []
                for t0 in range(0, NP, Bs):
                    chunks.append({k: v.clone() for k, v in eng.forward(fr[t0:t0 + Bs]).items() if hasattr(v, "shape") and v.shape[:1] == (Bs,)})
                    torch.cuda.synchronize()
                got = {k: torch.cat([q[k] for q in chunks]) for k in chunks[0]}
            else:
                fr = eng.inputs[0][P0:P0 + NP]
                eng.forward(slot=0)
                torch.cuda.synchronize()
                got = {k: v[P0:P0 + NP].clone() for k, v in eng.outputs().items() if hasattr(v, "shape") and v.shape[:1] == (Bs,)}
            parts = []
            for t0 in range(0, NP, NB):
                parts.append({k: v.clone() for k, v in ref.forward(fr[t0:t0 + NB]).items() if hasattr(v, "shape") and v.shape[:1] == (NB,)})
                torch.cuda.synchronize()
            want = {k: torch.cat([q[k] for q in parts]) for k in parts[0]}
            parity = {"bench_engine_vs_fp32_engine": engine_pair_stats(got, want, arch.nq),
                      "token_id_agreement": token_id_agreement(got, want, arch.nq),
                      # the benched engine's tracks scored AGAINST the fp32 engine's tracks as ground truth (100 = identical)
                      "agreement_hota": agreement_hota([tracks_of(got, b, cfg["W"], cfg["H"]) for b in range(NP)],
                                                       [tracks_of(want, b, cfg["W"], cfg["H"]) for b in range(NP)], device=dev),
                      "frames": NP, "first_frame": P0, "bench_engine": f"{dtype_name} B={Bs}", "reference_engine": f"f32 B={NB}",
                      "hota_note": "agreement_hota scores the benched engine's tracks AGAINST THE fp32 ENGINE'S TRACKS (100 = identical); HOTA against the "
                                   "synthetic scene's ground truth is ~0.005 for EVERY engine on these random-init weights (profiles/parity_r0*_c2*.json: "
                                   "0.0048-0.0056), so 'HOTA within 0.1 of the reference' holds by construction and is not a parity statement"}
            st_ = parity["bench_engine_vs_fp32_engine"]
            # TWO gates, reported separately:
            # (1) ABSOLUTE -- anchored to the spec, not to this build's own history.  fp32 engine: north_star's sentence itself (same
            #     query selection as the small-batch fp32 engine, boxes 1e-4, decoder output / scores 1e-3, no birth flipped, ids
            #     equal; the fp32 engine is held to the CPU oracle below).  16-bit engines: never further from fp32 than the
            #     ARITHMETIC TYPE itself -- the oracle executed in the same 16-bit type by eager torch on the first 8 frames of this
            #     window, in this run (the reference's own `half` switch, engine/predictor.py:131; filled in by the oracle leg).
            if dtype_name in ("f32", "f32x3"):
                ab = {"box_matched": 1e-4, "hs_matched": 1e-3, "score_matched": 1e-3, "birth_flip_frac_of_active": 0.0}
                # ids: the exact engine runs the SAME kernels at both batch sizes -> the id arrays are equal as arrays.  Split precision
                # moves a score by ~1e-5, so on these free-running stream frames (no top-k margins: two exact fp32 implementations
                # disagree on the ORDER of near-tied tokens there too) the bar is every token carrying the same
                # id in both runs; the order-exact comparison on the margin fixtures is the oracle leg below and tests/
                ids_ok = st_["ids_equal"] if dtype_name == "f32" else parity["token_id_agreement"]["tokens_id_equal_frac"] == 1.0
                parity["absolute"] = {"kind": "north_star sentence vs the small-batch fp32 engine", "bars": ab,
                                      "ok": bool(st_["box_max_err_matched"] <= ab["box_matched"] and st_["hs_max_err_matched"] <= ab["hs_matched"]
                                                 and st_["score_max_err_matched"] <= ab["score_matched"] and st_["births_flipped"] == 0
                                                 and ids_ok and n_masked == 0)}
            else:
                parity["absolute"] = {"kind": "benched engine vs eager torch in the same 16-bit type (same run, first 8 frames of the window)",
                                      "ok": None, "note": "evaluated below"}
            # (2) REGRESSION -- 1.5 x the committed measurement of THIS window (deterministic: same frames, same kernels, no atomics on
            #     the forward path), so that a 2 x regression of the 16-bit path fails.  It says nothing about closeness to the reference.
            #                                       box      hs     score   births / active        (profiles/r04_f_bench_*.json)
            measured = {("c2", "bf16"): (4.08e-3, 0.673, 0.147, 0.0744), ("c2", "f16"): (1.07e-3, 0.115, 0.0256, 0.0153),
                        ("c4", "bf16"): (4.89e-3, 0.638, 0.0650, 0.0293)}.get((cfg_name, dtype_name))
            if small or NP != 32 or (B // max(1, seq_per_gpu)) < P0 + NP:
                # the committed windows are frames 8..39 of ONE sequence through the bench-scale plan: a small-batch engine runs the classic
                # plan, a --batch below 40 sees a shorter window (a birth-flip RATIO over 8 frames is not the one over 32) and a batch of
                # several sequences (C5, 8 frames of each) a window of other frames
                measured = None
            if measured is not None:
                rb = tuple(round(1.5 * v, 5) for v in measured)
                parity["regression"] = {"kind": "1.5 x this window's committed measurement (profiles/r04_f_bench_*.json)",
                                        "bars": {"box_matched": rb[0], "hs_matched": rb[1], "score_matched": rb[2], "birth_flip_frac_of_active": rb[3]},
                                        "ok": bool(st_["box_max_err_matched"] <= rb[0] and st_["hs_max_err_matched"] <= rb[1]
                                                   and st_["score_max_err_matched"] <= rb[2] and st_["birth_flip_frac_of_active"] <= rb[3])}
            else:
                parity["regression"] = {"kind": "no committed measurement of this window", "ok": True}
            parity["sane"] = bool(n_masked == 0 and bool(torch.isfinite(got["boxes"]).all()) and st_["topk_overlap"] >= 0.8)
            if a.config == "full":
                # a TIMING configuration: its score heads were calibrated at another resolution and it has no committed window
                parity["gated"] = "sanity + the absolute gate (no committed window for this configuration)"

            def parity_ok():
                ab_ok = parity["absolute"]["ok"]             # (None only between here and the yardstick a few lines below)
                return bool(parity["sane"] and parity["regression"]["ok"] and (ab_ok is None or ab_ok))
            parity["ok"] = parity_ok()

            def yardstick(run16):
                """16-bit engines: eager torch in the same type on the first 8 frames of the window, both against the fp32 engine."""
                if dtype_name in ("f32", "f32x3"):
                    return None
                NY = min(8, NP)
                sl = lambda d: {k: d[k][:NY] for k in ("topk_ind", "boxes", "scores", "obj_idxes", "hs")}
                ye = run16(fr[:NY], dtype)
                se, sy = engine_pair_stats(sl(got), sl(want), arch.nq), engine_pair_stats(ye, sl(want), arch.nq)
                keys = ("box_max_err_matched", "hs_max_err_matched", "score_max_err_matched", "births_flipped", "birth_flip_frac_of_active", "topk_overlap")
                ok = bool(se["births_flipped"] <= sy["births_flipped"] + 1 and se["box_max_err_matched"] <= sy["box_max_err_matched"] + 1e-6
                          and se["hs_max_err_matched"] <= sy["hs_max_err_matched"] + 1e-6
                          and se["score_max_err_matched"] <= sy["score_max_err_matched"] + 1e-6
                          and se["topk_overlap"] >= sy["topk_overlap"] - max(0.005, 1.5 / arch.nq))
                return {"frames": NY, "engine": {k: se[k] for k in keys}, "eager_torch_same_dtype": {k: sy[k] for k in keys}, "ok": ok}
            NPc = NB
            # the engine held to the CPU oracle on the oracle's own (margin-fixture) frames: the small-batch fp32 engine, or -- when the
            # benched engine multiplies in split precision -- a small-batch engine of THAT kind
            chk = ref if dtype_name != "f32x3" else TrackEngine(arch, sd, cfg["H"], cfg["W"], batch=NB, dtype=torch.float32, device=dev, split_f16=True)

            def engine_check(keep):
                """fp32 engine vs the CPU oracle on the oracle's own frames: logits <= 1e-3, ids exact (given the same selection)."""
                import numpy as np
                from oracle import track_oracle as O
                u8 = torch.from_numpy(np.concatenate([k[0] for k in keep] * (NPc // len(keep) + 1))[:NPc]).to(dev)
                o = {k: v.clone() for k, v in chk.forward(u8).items()}
                torch.cuda.synchronize()
                res = {"frames": len(keep), "engine": "f32 split_f16" if dtype_name == "f32x3" else "f32", "logits_max_err": 0.0, "topk_equal": True, "ids_exact": True}
                for i, (_, r, ids) in enumerate(keep):
                    tk = o["topk_ind"][i].cpu().long()
                    same = bool(torch.equal(tk, r["topk_ind"][0]))
                    res["topk_equal"] &= same
                    if same:
                        res["logits_max_err"] = max(res["logits_max_err"], float((o["logits"][i].cpu() - r["dec_scores"][0]).abs().max()))
                        res["ids_exact"] &= bool(torch.equal(o["obj_idxes"][i].cpu(), ids))
                res["ok"] = bool(res["logits_max_err"] <= 1e-3 and res["ids_exact"] and res["topk_equal"])
                return res
        else:
            engine_check = yardstick = None
        if a.temporal and not a.no_parity:
            # carried-query mode: the benched engine and an fp32 engine of the same weights run the first GATE_T frames of the first 4
            # sequences from a reset.  Reported AND gated (VERDICT r3 #3a): the tracks of the benched engine scored against the fp32
            # engine's tracks as ground truth -- HOTA / DetA / AssA by the published definition (ids carry across frames here, so the
            # association half means something) --, `n_overflow` (active rows beyond the slots: dropped, never silent), and the
            # detect-query rows of the first 3 steps by token as before.  (Spec parity only, DESIGN.md section 7: the reference's
            # carried branch cannot run; the fp32 engine itself is held to oracle/temporal_oracle.py by tests/test_gpu_temporal.py.)
            GATE_T = 24
            log(f"parity gate (temporal): benched engine vs fp32 temporal engine, {GATE_T} steps from reset")
            import numpy as np
            from mo_yolo_amd.parity import _xyxy, agreement_hota, engine_pair_stats
            NB, nm = min(4, B), a.temporal
            ref = TrackEngine(arch, sd, cfg["H"], cfg["W"], batch=NB, dtype=torch.float32, device=dev, temporal=a.temporal)
            eng.reset_sequence()
            steps = []
            worst = {"box_max_err_matched": 0.0, "hs_max_err_matched": 0.0, "score_max_err_matched": 0.0}
            flips = active = 0
            trk_g, trk_w = [[] for _ in range(NB)], [[] for _ in range(NB)]
            over_g = over_w = 0
            reps = (B + NB - 1) // NB

            def trk_of(o, b):
                ids, bx = o["obj_idxes"][b].cpu(), o["boxes"][b].float().cpu()
                act = ids >= 0
                return _xyxy(bx[act], cfg["W"], cfg["H"]).numpy().astype("float32"), ids[act].numpy().astype("int64")

            for k in range(GATE_T):
                fr4 = torch.from_numpy(np.concatenate([s_.frames(k, 1) for s_ in seqs[:NB]])).to(dev)
                eng.forward(fr4.repeat(reps, 1, 1, 1)[:B].contiguous(), slot=0)      # sequences 0..3 in the first rows of the benched batch
                torch.cuda.synchronize()
                go = {kk: v[:NB].clone() for kk, v in eng.outputs().items() if hasattr(v, "shape") and v.shape[:1] == (B,)}
                wo = {kk: v.clone() for kk, v in ref.forward(fr4).items() if hasattr(v, "shape") and v.shape[:1] == (NB,)}
                torch.cuda.synchronize()
                over_g += int(go["n_overflow"].sum()); over_w += int(wo["n_overflow"].sum())
                for b in range(NB):
                    trk_g[b].append(trk_of(go, b)); trk_w[b].append(trk_of(wo, b))
                if k < 3:
                    cut = lambda o: dict(topk_ind=o["topk_ind"], boxes=o["boxes"][:, nm:], scores=o["scores"][:, nm:],
                                         obj_idxes=o["obj_idxes"][:, nm:], hs=o["hs"][:, nm:])
                    st = engine_pair_stats(cut(go), cut(wo), arch.nq)
                    for kk in worst:
                        worst[kk] = max(worst[kk], st[kk])
                    flips += st["births_flipped"]; active += st["active_rows_reference"]
                    steps.append({"step": k, "topk_overlap": st["topk_overlap"], "births_flipped": st["births_flipped"],
                                  "live_tracks": [int(v) for v in go["n_tracks"]], "live_tracks_fp32": [int(v) for v in wo["n_tracks"]]})
            eng.reset_sequence()
            ag = [agreement_hota(trk_g[b], trk_w[b], device=dev)["published"] for b in range(NB)]
            ag_min = {m: round(min(x[m] for x in ag), 3) for m in ("HOTA", "DetA", "AssA")}
            bars = {"f32": (1e-4, 1e-3, 1e-3, 0.0), "f16": (5e-3, 0.6, 0.16, 0.03), "bf16": (9e-3, 1.3, 0.4, 0.1)}[dtype_name]
            if (cfg_name, dtype_name, a.temporal, B) == ("c2", "bf16", 100, 32):     # 1.5 x profiles/r03_f_bench_c2_temporal100.json
                bars = (4.5e-3, 0.5, 0.184, 0.052)
            # agreement bars (published HOTA / DetA / AssA, minimum over the 4 sequences): 100 - 1.5 x (100 - the committed measurement)
            hota_bars = {"f32": (99.9, 99.9, 99.9), "f16": (85.0, 80.0, 88.0), "bf16": (65.0, 55.0, 72.0)}[dtype_name]
            if (cfg_name, dtype_name, a.temporal, B) == ("c2", "bf16", 100, 32):     # measured (deterministic): min over 4 sequences 84.6 / 77.9 / 90.8
                hota_bars = (76.9, 66.9, 86.2)
            frac = flips / max(1, active)
            parity = {"temporal": True, "steps": steps, "detect_rows_vs_fp32": dict(worst, births_flipped=flips, active_rows_reference=active,
                                                                                    birth_flip_frac_of_active=round(frac, 5)),
                      "agreement_hota_vs_fp32_temporal_engine": {"frames": GATE_T, "sequences": NB, "per_sequence": ag, "min": ag_min,
                                                                 "definition": "published HOTA; the benched engine's tracks scored against the fp32 temporal engine's tracks as ground truth"},
                      "n_overflow": {"benched_engine": over_g, "fp32_engine": over_w, "slots": nm,
                                     "note": "active rows beyond the track slots are dropped and counted (this throughput configuration saturates its slots: DESIGN.md section 2.4)"},
                      "bars": {"box_matched": bars[0], "hs_matched": bars[1], "score_matched": bars[2], "birth_flip_frac_of_active": bars[3],
                               "agreement_HOTA_min": hota_bars[0], "agreement_DetA_min": hota_bars[1], "agreement_AssA_min": hota_bars[2]},
                      "bench_engine": f"{dtype_name} B={B} temporal={a.temporal}", "reference_engine": f"f32 B={NB} temporal={a.temporal}"}
            parity["ok"] = bool(worst["box_max_err_matched"] <= bars[0] and worst["hs_max_err_matched"] <= bars[1]
                                and worst["score_max_err_matched"] <= bars[2] and frac <= bars[3]
                                and ag_min["HOTA"] >= hota_bars[0] and ag_min["DetA"] >= hota_bars[1] and ag_min["AssA"] >= hota_bars[2])

        if parity is not None and yardstick is not None and dtype_name not in ("f32", "f32x3"):
            # the absolute gate of a 16-bit engine needs only the GPU and 8 frames -- it runs whether or not the CPU baseline
            # does (N > 1, --no-cpu-baseline), and a yardstick that raises FAILS the gate instead of leaving it unevaluated
            log("absolute gate: eager torch in the same 16-bit type on the first 8 frames of the window")
            try:
                yard = yardstick(eager_same_dtype_runner(sd, arch))
                parity["absolute"].update(ok=yard["ok"], yardstick=yard)
                parity["absolute"].pop("note", None)
            except Exception as e:
                parity["absolute"].update(ok=False, error=repr(e)[:300])
                parity["absolute"].pop("note", None)
            parity["ok"] = parity_ok()
        if world == 1 and not a.no_cpu_baseline:
            log("cpu baseline (oracle on the host cores)")
            try:
                cpu, par_cpu = cpu_baseline(cfg, arch, sd, a.cpu_frames, engine_check)
                if parity is not None and par_cpu is not None:
                    parity["fp32_engine_vs_cpu_oracle"] = par_cpu
                if parity is not None and not a.temporal:
                    parity["ok"] = bool(parity_ok() and (par_cpu is None or par_cpu["ok"]))
                elif parity is not None and par_cpu is not None:
                    parity["ok"] = bool(parity["ok"] and par_cpu["ok"])
            except Exception as e:  # the baseline must never lose the measured line
                cpu = {"error": repr(e)}

    n_launches = eng.num_launches if (rank == 0 and not a.dry_run) else 0

    # ---- self-check OUTSIDE the timed region (round 5): the benched plan replayed twice on the same input slot must
    # reproduce itself bit for bit (outputs, value planes, all token scores, every layer view), and its value planes must equal the
    # tiled kernel's on a sub-batch (the weight-stationary forms are bit-identical to it by construction) -- mo_yolo_amd/stress.py
    selfcheck = None
    if rank == 0 and not a.dry_run and not a.no_selfcheck and not a.predictor and not a.temporal:
        log("self-check: two replays of the benched plan bit for bit; value planes vs the tiled kernel")
        from mo_yolo_amd import stress as S_
        try:
            bad, _first = S_.plan_determinism(eng, passes=2, slot=0, eager=a.no_graph)
            del _first
            vp = S_.value_planes_vs_tiled(eng)
            selfcheck = {"replays_bit_identical": not bad, "value_planes_equal_tiled": bool(vp["equal"]), "ok": bool(not bad and vp["equal"]),
                         "detail": {"mismatches": bad[:8], "value_planes_vs_tiled": vp}}
        except Exception as e:  # pragma: no cover
            selfcheck = {"ok": False, "error": repr(e)[:300]}

    legs = {}
    if (rank == 0 and world == 1 and not a.dry_run and a.extra_legs and a.config == "c2" and not a.temporal
            and a.dtype is None and a.batch is None and not a.from_host and not a.predictor and not a.resize_from):
        # the other BASELINE.json configurations and the other ways frames reach the engine: child processes of this one (the
        # parent's plan is released first), own parity gates included; every leg is printed as its OWN short line, before the final one
        import gc
        import subprocess
        pipe = eng = None
        gc.collect()
        torch.cuda.empty_cache()
        leg_list = (("c4_bf16", ["--config", "c4"], 20), ("c5_f16", ["--config", "c5"], 20),
                    # C5 as BASELINE.json words it -- a hipGraph-captured PER-FRAME step over 4 live sequences -- and the carried-query mode at
                    # 4 sequences: one replay = one frame of each sequence, device synchronised after every step
                    ("c5_f16_b4_latency", ["--config", "c5", "--batch", "4", "--streams", "1", "--latency"], 300),
                    ("c2_bf16_temporal100_b4_latency", ["--temporal", "100", "--batch", "4", "--latency"], 300),
                    ("c2_bf16_temporal100", ["--temporal", "100", "--batch", "32"], 20),
                    ("c2_f32", ["--dtype", "f32"], 10),                  # the engine that meets "bit-exact ids" (fp32)
                    ("c2_f32x3", ["--dtype", "f32x3"], 10),              # ... and the same engine with split-fp16 matrix products
                    ("full_bf16", ["--config", "full"], 20),             # the reference's own model scale (yolo_track.yaml 1.0 / 1.0)
                    ("c2_bf16_from_host", ["--from-host"], 20),
                    ("c2_bf16_from_host_1080p_resize", ["--from-host", "--resize-from", "1080x1920"], 20),
                    ("c2_bf16_predictor", ["--predictor"], 5),
                    ("c2_bf16_predictor_pinned_source", ["--predictor", "--pinned-source"], 5),
                    ("c2_bf16_sustained_200_steps", [], 200))
        for name, flags, nsteps in leg_list:
            log(f"extra leg {name}")
            side = os.path.join(os.path.dirname(full_path(a)), f"bench_full_{name}.json")
            cmd = [sys.executable, os.path.abspath(__file__), *flags, "--steps", str(nsteps), "--warmup", "2", "--no-cpu-baseline", "--no-live-traffic",
                   "--full-out", side] + ([] if name in ("c2_f32", "c2_f32x3", "full_bf16") else ["--no-launch-table"]) \
                  + (["--no-parity", "--no-selfcheck"] if "sustained" in name else []) + (["--no-selfcheck"] if "latency" in name else [])
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
                d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
                leg = {"leg": name, "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"], "dtype": d["dtype"],
                       "latency_ms": (d.get("config") or {}).get("latency_ms"), "launches_per_step": (d.get("config") or {}).get("launches_per_step"),
                       "parity_ok": (d.get("parity") or {}).get("ok"), "selfcheck_ok": (d.get("selfcheck") or {}).get("ok"), "rc": r.returncode,
                       "full": d.get("full")}
                if d.get("roofline"):
                    leg["roofline_frac"] = d["roofline"].get("frac")
                cname = "c4" if name.startswith("c4") else "c2"
                if "temporal" not in name and not name.startswith("full") and "latency" not in name:    # SURVEY §8(d) figure: FPS x algorithmic bytes per frame / HBM peak
                    leg["roofline_frac_survey_8d"] = round(d["value"] * ALG_BYTES_FRAME[cname] * (2 if d["dtype"] in ("f32", "f32x3") else 1) / (HBM_PEAK_GBS * 1e9), 4)
                if name in ("c2_bf16_from_host", "c2_bf16_predictor", "c2_bf16_predictor_pinned_source", "c2_bf16_sustained_200_steps"):
                    leg["vs_resident_headline"] = round(d["value"] / fps, 4)
            except Exception as e:  # a failing leg must not lose the headline line
                leg = {"leg": name, "error": repr(e)[:300]}
            legs[name] = leg
            print(json.dumps(leg), flush=True)

    if rank == 0:
        if a.dry_run:
            workload = f"DRY RUN (no GPU): synthetic step, {frames_step} frames/step/rank"
            launches = 0
        else:
            per_seq = B // max(1, len(my_seqs))
            mode = (f"temporal mode, {B} sequences in lockstep, {a.temporal} track slots" if a.temporal else
                    f"SMALL-BATCH latency leg: one replay = ONE frame of each of {B} live sequences, device synchronised after every step" if a.latency else
                    f"{B} frames/step/GPU as {S} sub-batches on {S} HIP streams = {per_seq} consecutive frames of each sequence, every frame inside "
                    f"its 600-frame sequence")
            if a.temporal and a.latency:
                mode += ", device synchronised after every step (latency leg)"
            scale = ("yolo_track.yaml at depth 1.0 / width 1.0 (46 M parameters)" if a.config == "full" else "YOLOv8 s-scale")
            feed = ("uint8 frames resident in HBM (input slots, no per-step copy)" if not (a.from_host or a.predictor) else
                    "uint8 frames fed from PINNED HOST memory inside the timed region (copy stream, overlapped)" if a.from_host else
                    "uint8 frames handed to TrackPredictor.__call__ as pageable host arrays inside the timed region")
            if a.predictor:
                mode = f"{frames_step} frames per call in chunks of {B}, {S} engine(s) taking the chunks in turn on {S} HIP stream(s)"
            workload = (f"{a.config.upper()}: {scale} backbone/neck + 6-layer MOTR decoder, {arch.nq} queries, {cfg['W']}x{cfg['H']}, "
                        f"{feed}, {mode}, {len(my_seqs)} sequence(s) per GPU, "
                        f"{'eager' if a.no_graph else 'hipGraph replay'}")
            launches = n_launches
        metric = {"c2": "frames/sec (whole node) on 1088x608 MOT17 streams", "c5": "frames/sec (whole node) on 1088x608 MOT17 streams",
                  "full": "frames/sec (whole node) on 1088x608 MOT17 streams",
                  "c4": "frames/sec (whole node) on 1920x1088 DanceTrack-shape streams"}[a.config]
        full = {
            "metric": metric, "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "none" if a.dry_run else dtype_name, "data": "synthetic",
            "config": dict({"workload": workload, "frames_per_step_per_gpu": frames_step, "streams": S, "graph": not a.no_graph,
                            "launches_per_step": launches, "weights": "seeded synthetic (fixture recipe, mo_yolo_amd/fixtures.py)",
                            "sequences_of_rank0": my_seqs, "control_backend": backend, "hip_visible_devices_rank0": visible,
                            "dry_run": bool(a.dry_run)}, **line_extra),
            "roofline": roof, "roofline_step": roof_step, "parity": parity, "selfcheck": selfcheck,
        }
        if cpu is not None:
            full["cpu_baseline"] = cpu
        if legs:
            full["legs"] = legs
        path = full_path(a)
        try:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            with open(path, "w") as f:
                json.dump(full, f)
        except OSError as e:  # pragma: no cover
            path = f"(not written: {e})"
        shown = os.path.relpath(path, ROOT) if os.path.abspath(path).startswith(ROOT + os.sep) else path
        print(json.dumps(compact_line(full, shown)), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if parity is not None and not parity.get("ok", True):
        print("bench.py: PARITY GATE FAILED: " + json.dumps(parity)[:4000], file=sys.stderr, flush=True)
        sys.exit(3)
    if selfcheck is not None and not selfcheck.get("ok", True):
        print("bench.py: SELF-CHECK FAILED: " + json.dumps(selfcheck)[:4000], file=sys.stderr, flush=True)
        sys.exit(4)


def full_path(a):
    if a.full_out:
        return os.path.abspath(a.full_out)
    d = os.path.join(ROOT, "gpurun_out")
    return os.path.join(d if os.path.isdir(d) else ROOT, "bench_full.json")


def _r(v, nd=4):
    return round(v, nd) if isinstance(v, float) else v


def compact_line(full: dict, path: str = "bench_full.json") -> dict:
    """The ONE line the driver parses: the contract's keys + `roofline`, `cpu_baseline`, `parity` and `selfcheck` as a
    few scalars each -- under 4 KB whatever the run measured (tests/test_host_logic.py).  Everything else is in the side file `full`."""
    keys = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    line = {k: full.get(k) for k in keys}
    c = full.get("config") or {}
    cfg = {k: c.get(k) for k in ("workload", "frames_per_step_per_gpu", "streams", "graph", "launches_per_step", "dry_run", "control_backend")}
    for k in ("mean_active_tracks", "latency_ms", "startup_s_rank0", "host_rss_mb_rank0"):
        if c.get(k) is not None:
            cfg[k] = c[k]
    cfg["workload"] = str(cfg["workload"])[:480]
    if c.get("hbm_allocated_gb") is not None:
        cfg["hbm_allocated_gb"] = c["hbm_allocated_gb"]
    for k in ("sequences_of_rank0", "rank_map", "ranks"):          # the N > 1 diagnostics: short per-rank records
        if k in c:
            v = c[k]
            if k == "ranks":
                v = [{kk: e.get(kk) for kk in ("rank", "dt_local_s", "fps_local", "hip_visible_devices", "sequences", "pci_bus_id", "startup_s", "host_rss_mb")} for e in v]
            cfg[k] = v
    line["config"] = cfg
    r = full.get("roofline")
    if r:
        line["roofline"] = {k: _r(r.get(k)) for k in ("bound", "kernel", "calls_per_pass", "avg_ms", "alg_bytes_per_launch", "achieved", "peak", "unit",
                                                      "frac", "traffic", "traffic_frac", "limiter", "traffic_source")
                            if k in r or k in ("traffic", "traffic_frac", "traffic_source")}
        if line["roofline"].get("traffic_source"):
            line["roofline"]["traffic_source"] = str(line["roofline"]["traffic_source"])[:120]
        line["roofline"]["kernel"] = str(line["roofline"].get("kernel"))[:80]
        if r.get("runner_up"):
            ru = r["runner_up"]
            line["roofline"]["runner_up"] = {"kernel": str(ru.get("kernel"))[:60], "avg_ms": ru.get("avg_ms"), "frac": ru.get("frac")}
    else:
        line["roofline"] = None
    rs = full.get("roofline_step")
    if rs:
        line["roofline_step"] = {k: rs.get(k) for k in ("frac", "traffic_frac", "sum_of_floors_frac", "sum_kernel_ms_eager") if k in rs}
    cb = full.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = ({k: cb.get(k) for k in ("value", "unit", "cores", "kind", "cores_available", "frames_per_call", "numeric_only_fps")} if "error" not in cb
                                else {"error": str(cb["error"])[:200]})
        if "sample" in cb:
            line["cpu_baseline"]["sample"] = str(cb["sample"])[:220]
    p = full.get("parity")
    if p:
        st = p.get("bench_engine_vs_fp32_engine") or {}
        tk = p.get("token_id_agreement") or {}
        ah = (p.get("agreement_hota") or {}).get("published") or {}
        q = {"ok": p.get("ok"), "frames": p.get("frames"), "vs": p.get("reference_engine"),
             "box_max_err": _r(st.get("box_max_err_matched"), 6), "hs_max_err": _r(st.get("hs_max_err_matched"), 5),
             "score_max_err": _r(st.get("score_max_err_matched"), 5), "birth_flip_frac": st.get("birth_flip_frac_of_active"),
             "ids_equal": st.get("ids_equal"), "topk_order_equal_frames": st.get("topk_order_equal_frames"),
             "tokens_id_equal_frac": tk.get("tokens_id_equal_frac"), "agreement_DetA": ah.get("DetA"), "agreement_HOTA": ah.get("HOTA")}
        if ah:
            q["hota_note"] = "agreement vs the fp32 engine's tracks; HOTA vs the synthetic GT is ~0.005 for every engine on random-init weights (no parity information)"
        ab = p.get("absolute") or {}
        q["absolute_ok"] = ab.get("ok")
        if ab.get("error"):
            q["absolute_error"] = str(ab["error"])[:120]
        y = ab.get("yardstick")
        if y:
            q["births_flipped_engine_vs_eager_same_dtype"] = [y["engine"].get("births_flipped"), y["eager_torch_same_dtype"].get("births_flipped")]
        q["regression_ok"] = (p.get("regression") or {}).get("ok")
        o = p.get("fp32_engine_vs_cpu_oracle")
        if o:
            q["fp32_engine_vs_cpu_oracle"] = {"ok": o.get("ok"), "logits_max_err": _r(o.get("logits_max_err"), 7), "ids_exact": o.get("ids_exact"),
                                              "topk_equal": o.get("topk_equal")}
        if p.get("temporal"):
            q["temporal"] = True
            q["agreement_min"] = (p.get("agreement_hota_vs_fp32_temporal_engine") or {}).get("min")
            q["n_overflow"] = {k: v for k, v in (p.get("n_overflow") or {}).items() if k != "note"}
        line["parity"] = {k: v for k, v in q.items() if v is not None or k in ("ok", "absolute_ok")}
    else:
        line["parity"] = None
    sc = full.get("selfcheck")
    if sc:
        line["selfcheck"] = {k: sc.get(k) for k in ("ok", "replays_bit_identical", "value_planes_equal_tiled", "error") if k in sc}
    if full.get("legs"):
        line["legs"] = {k: v.get("value", "error") for k, v in full["legs"].items()}
    line["full"] = path
    for drop in (("legs",), ("roofline_step",), ("config", "rank_map"), ("roofline", "runner_up"), ("config", "ranks")):   # never reached by the runs
        if len(json.dumps(line)) < 4000:                                                                      # measured so far: a guarantee
            break
        d = line
        for k in drop[:-1]:
            d = d.get(k) or {}
        d.pop(drop[-1], None)
    return line


if __name__ == "__main__":
    main()

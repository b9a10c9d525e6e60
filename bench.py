#!/usr/bin/env python3
"""Headline benchmark: whole-job frames/s of the per-frame tracking step on synthetic 1088x608 MOT17-shape streams
(BASELINE.json metric; config C2 at N=1, C3 = one sequence shard per GPU at N>1, no collective on the data path).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path (fused preprocess -> backbone/neck -> decoder -> ID assignment + predictor rows) over
one batch of `--batch` frames already resident in HBM (the engines' input slots: a step copies no frame bytes), replayed as
hipGraphs.  Rank 0 prints ONE JSON line.  Other workloads: `--config c4` (1920x1088, 500 queries), `--config c5` (fp16,
4 sequences batched per GPU), `--temporal N` (carried track queries, batch element = sequence), `--dtype f16|f32`.
`--dry-run --backend gloo` runs the rank -> sequence / barrier / MAX-reduce / rank-0-JSON control flow without a GPU
(CPU test of the N > 1 path).
"""
import argparse
import json
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
                    help="frames per step per GPU (default 576 at C2: 288 per sub-batch engine is the largest whose buffers stay "
                         "inside 2 GiB descriptors; 128 at C4; temporal mode: sequences per GPU)")
    ap.add_argument("--config", default="c2", choices=["c2", "c4", "c5", "full"],
                    help="full = yolo_track.yaml at its own depth 1.0 / width 1.0 (the scale the reference's entry script trains, "
                         "start_train.py:11) at 1088x608, 300 queries")
    ap.add_argument("--dtype", default=None, choices=["bf16", "f16", "f32"])
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
    ap.add_argument("--cpu-frames", type=int, default=20)
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
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="default C2 run at N=1 appends the other workloads (child processes, 20 steps each) under `extra`")
    return ap.parse_args(argv)


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


def cpu_baseline(cfg, arch, sd, n_frames, engine_check=None):
    """Oracle (a port of the reference's eager path, verified against it in the build container) timed on this box's host
    cores, BASELINE.md §4: fp32, all cores, 3 warm-ups + n_frames; FPS for (i) the numeric graph only and (ii) including the
    reference-faithful Python state machine.  `engine_check(frames_u8, oracle_result)` is the parity gate of BASELINE.md §5:
    the oracle is the checker of the fp32 engine on the very frames it is timed on."""
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
    budget_s = 45.0                                     # bounded sample: the default bench run must finish within minutes
    with torch.no_grad():
        # eager batch-1 ops stop scaling beyond ~16 threads and collapse with hundreds (measured on the MI355X box: 0.095 s/frame
        # with 16 threads, 123.6 s/frame with all 256): time one frame with 16 and with 32 threads, keep the faster setting and
        # report the thread count actually used beside the cores available
        x0 = to_network_input(seq.frames(0, 1))
        best = None
        for c in sorted({min(cores_all, 16), min(cores_all, 32)}):
            torch.set_num_threads(c)
            O.forward(x0, sd, arch)
            t0 = time.perf_counter()
            O.forward(x0, sd, arch)
            dtc = time.perf_counter() - t0
            print(f"[cpu_baseline] {c} threads: {dtc:.3f} s/frame", file=sys.stderr, flush=True)
            if best is None or dtc < best[0]:
                best = (dtc, c)
        cores = best[1]
        torch.set_num_threads(cores)
        O.forward(x0, sd, arch)                         # third warm-up at the chosen setting
        keep = []
        done = 0
        t_begin = time.perf_counter()
        for i in range(n_frames):
            u8 = seq.frames(i, 1)
            x = to_network_input(u8)
            t0 = time.perf_counter()
            r = O.forward(x, sd, arch)
            t1 = time.perf_counter()
            scores = r["dec_scores"][0].sigmoid().max(-1).values
            ids, _, _ = O.assign_ids_loop(scores)                       # host loop as shipped (head.py:1232-1243)
            ids = torch.tensor(ids)
            O.tracker_update_copy(scores.tolist(), r["dec_bboxes"][0].numpy(), ids.tolist())
            O.postprocess(r["y"][0], r["dec_scores"][0], ids, 0.25, orig_hw=(cfg["H"], cfg["W"]))
            t2 = time.perf_counter()
            t_num += t1 - t0
            t_state += t2 - t1
            done += 1
            if i < 2:
                keep.append((u8, r, ids))
            if done % 5 == 0:
                print(f"[cpu_baseline] {done}/{n_frames} frames", file=sys.stderr, flush=True)
            if time.perf_counter() - t_begin > budget_s and done >= 5:
                break
        n_frames = done
        if engine_check is not None:
            parity = engine_check(keep)
    out = {"value": n_frames / (t_num + t_state), "unit": "frames/s", "cores": cores, "kind": "port",
           "numeric_only_fps": n_frames / t_num,
           "cores_available": cores_all,
           "sample": f"{n_frames} frames (after 3 warm-ups) of the same synthetic stream, batch 1, fp32 eager torch-CPU oracle; "
                     f"value includes the reference-faithful host state machine, numeric_only_fps excludes it"}
    return out, parity


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
    a = parse(argv)
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    visible = None
    # (the dry run pins too -- it only writes the environment variable -- so the exact rank -> device map is observable without a GPU)
    visible = pin_device(0 if a.rehearse_one_gpu else local, a.rehearse_one_gpu)
    cpus = pin_cpus(local, int(os.environ.get("LOCAL_WORLD_SIZE", world)))      # before anything touches HIP / starts threads

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
    else:
        # (fp32 buffers are twice the size: 96 frames per engine; the full-width model's widest buffer -- the first C2f's
        #  [y0 | y1 | y2 | y3 | y4] at 152 x 272 x 320 channels -- allows 81 per engine inside 2 GiB descriptors; 76 makes the 256-row tiles of
        #  the 38 x 68 level fill exactly three rounds of 256 CUs: +1.3 % over 64)
        B = a.batch or {"c4": 128, "full_c2": 152}.get(cfg_name, 192 if dtype_name == "f32" else 576)
        if a.predictor:
            B = a.batch or 288
        S = max(1, a.streams if a.streams is not None else 2)
    if B % S or (not a.temporal and B % seq_per_gpu):
        raise SystemExit("--batch must be a multiple of --streams and of the sequences per GPU")
    # sequence shard of this rank (sequence i -> rank i mod N, no cross-GPU term)
    my_seqs = shard.sequences_for_rank(world * seq_per_gpu, rank, world)

    def barrier():
        if world > 1:
            dist.barrier()
        if not a.dry_run:
            torch.cuda.synchronize()

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
        dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[dtype_name]
        seqs = [SyntheticSequence(sid, cfg["H"], cfg["W"], cfg["style"]) for sid in my_seqs]
        n_slots = 3

        def batch_frames(i):
            """Step i's frames: per-frame mode -> B/seq_per_gpu consecutive frames of every sequence of this rank;
            temporal mode -> frame i of each of the B sequences."""
            import numpy as np
            if a.temporal:
                return torch.from_numpy(np.concatenate([s.frames(i, 1) for s in seqs])).to(dev)
            per = B // len(seqs)
            # the synthetic sequences are 600 frames long by design (the rectangles have left the scene soon after):
            # slot i starts at frame i * per while that stays inside the sequence, else the slots are windows 8 frames apart
            t0 = i * per if (i + 1) * per <= 600 else 8 * i
            return torch.from_numpy(np.concatenate([s.frames(t0, per) for s in seqs])).to(dev)

        if a.temporal:
            eng = TrackEngine(arch, sd, cfg["H"], cfg["W"], batch=B, dtype=dtype, device=dev, temporal=a.temporal, n_inputs=n_slots)
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
            B = n_chunks * B                                     # frames per timed step
        else:
            pipe = StreamedEngines(arch, sd, cfg["H"], cfg["W"], batch=B, streams=S, graph=not a.no_graph, dtype=dtype, device=dev,
                                   n_inputs=n_slots)
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
    for i in range(a.warmup):
        step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(i)
    barrier()
    dt_local = time.perf_counter() - t0
    dt = shard.max_over_ranks(dt_local, device=(torch.device("cuda", 0) if backend == "nccl" else None))
    fps = shard.whole_job_fps(B * a.steps, dt, world)
    if world > 1:
        # the first hardware multi-GPU run must be diagnosable -- every rank's own time and device, not only the MAX
        me = {"rank": rank, "local_rank": local, "dt_local_s": round(dt_local, 6), "fps_local": round(B * a.steps / dt_local, 2),
              "hip_visible_devices": visible, "cpus": cpus, "sequences": my_seqs, "host": os.uname().nodename}
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
            dom = max(range(nL), key=lambda i: per[i])
            m = eng.meta[dom]
            ach = m["bytes"] / (per[dom] * 1e-3) / 1e9 if m["bytes"] else 0.0
            prof = {}
            tp = os.path.join(ROOT, "profiles", "traffic_by_launch.json")
            if os.path.exists(tp):
                prof = json.load(open(tp))
            traffic = prof.get("launches", prof).get(m["name"], {}).get("hbm_bytes") if dtype_name == "bf16" else None   # (measured on the bf16 plan)
            roof = {"bound": "hbm", "kernel": m["name"], "launch_index": dom, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "avg_ms": round(per[dom], 4), "share_of_step": round(per[dom] / sum(per), 4),
                    "alg_bytes_per_launch": m["bytes"], "tflops": round(m["flops"] / (per[dom] * 1e-3) / 1e12, 2)}
            # Round 4: the value projection (1.6-1.9 ms, HBM-bound) and the fused stem (1.62-1.67 ms, bound by vector-instruction issue:
            # SiLU + the uint8 fragment build, 15 % matrix-pipe busy -- profiles/r04_b_pmc_all_kernels_b288_1stream.txt) are within a few per
            # cent of each other, so which one is "dominant" differs by device: the runner-up is always reported beside it
            order = sorted(range(nL), key=lambda i: -per[i])
            if len(order) > 1:
                m2 = eng.meta[order[1]]
                ach2 = m2["bytes"] / (per[order[1]] * 1e-3) / 1e9 if m2["bytes"] else 0.0
                tr2 = prof.get("launches", prof).get(m2["name"], {}).get("hbm_bytes") if dtype_name == "bf16" else None
                roof["runner_up"] = {"kernel": m2["name"], "launch_index": order[1], "avg_ms": round(per[order[1]], 4), "achieved": round(ach2, 1),
                                     "frac": round(ach2 / HBM_PEAK_GBS, 4), "traffic": tr2, "alg_bytes_per_launch": m2["bytes"],
                                     "tflops": round(m2["flops"] / (per[order[1]] * 1e-3) / 1e12, 2)}
            if m["name"].startswith("stem"):
                roof["note"] = ("fused preprocess + stem + conv1: bound by vector-instruction issue (SiLU at 28 cycles per value, uint8 fragment build), "
                                "neither the HBM nor the matrix roof is near; the HBM-bound value projection is `runner_up`")
            if dtype_name == "f32":
                # the exact-fp32 engine multiplies on v_mfma_f32_16x16x4_f32: 1/16 of the bf16 matrix rate, the same as the fp32
                # vector rate (MI355X_MICROARCH.md: 157.3 TFLOP/s spec, 155 measured) -- its dominant launch is bound by THAT
                tf = m["flops"] / (per[dom] * 1e-3) / 1e12
                roof.update(bound="mfma", achieved=round(tf, 2), peak=MFMA_F32_PEAK_TFLOPS, unit="TFLOP/s",
                            frac=round(tf / MFMA_F32_PEAK_TFLOPS, 4), hbm_gbs=round(ach, 1))
            ms_step = dt / a.steps * 1e3
            n_eng = 1 if pipe is None else len(pipe.engines)
            plan_bytes = sum(mm["bytes"] for mm in eng.meta) * n_eng
            roof_step = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "launches": nL, "sum_kernel_ms_eager": round(sum(per), 3),
                         # (b) what the plan's launches move by their own fused-op accounting (every launch kind has a byte count)
                         "plan_bytes_per_step": plan_bytes,
                         "plan_frac": round(plan_bytes / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            if cfg_name in ALG_BYTES_FRAME and not a.temporal:
                # (a) SURVEY §8(d) convention: layer-wise algorithmic bytes of the reference's op list (the figure the 60 % target
                # is stated in; it charges value_proj's input six times and enc_output over all S tokens, which this plan avoids)
                bytes_step = (ALG_BYTES_FRAME[cfg_name] - ALG_WEIGHT_BYTES) * B + ALG_WEIGHT_BYTES * (1 if pipe is None else n_eng)
                if dtype_name == "f32":
                    bytes_step *= 2
                ach_s = bytes_step / (ms_step * 1e-3) / 1e9
                roof_step.update(achieved=round(ach_s, 1), frac=round(ach_s / HBM_PEAK_GBS, 4), alg_bytes_per_step=bytes_step)
                # SURVEY §8(d) as written charges the 25.7 MB of weights to EVERY frame (FPS x B_alg / peak: 11.6 k FPS = 0.60 at C2);
                # `frac` above charges them once per launch (they are read once per sub-batch), the stricter figure
                roof_step["frac_survey_8d_per_frame_weights"] = round(ALG_BYTES_FRAME[cfg_name] * (2 if dtype_name == "f32" else 1) * B
                                                                      / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            # (c) measured HBM traffic of the plan (PMC passes committed under profiles/, tools/pmc_traffic.sh): bytes per frame
            tpf = prof.get("step_total", {}).get(f"{cfg_name}_{dtype_name}", {}).get("hbm_bytes_per_frame")
            if tpf and not a.temporal:                           # (measured on the per-frame plan)
                tb = tpf * B
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
            mfma_peak = MFMA_F32_PEAK_TFLOPS if dtype_name == "f32" else MFMA_PEAK_TFLOPS
            floors = [max(mm["bytes"] / (HBM_PEAK_GBS * 1e9), mm["flops"] / (mfma_peak * 1e12)) * 1e3 for mm in eng.meta]
            roof_step["sum_of_launch_floors_ms"] = round(sum(floors), 3)
            roof_step["sum_of_floors_frac"] = round(sum(floors) / max(sum(per), 1e-9), 4)
            # the same against what THIS device sustains: the copy rate measured above (a stream that reads and writes does not get the
            # fill rate: DESIGN.md round-3 item 12) and the sustained MFMA rate -- informative, never `roofline.frac`
            cp = roof_step.get("copy_peak_gbs_measured")
            if isinstance(cp, float) and cp > 0:
                msus = 155.0 if dtype_name == "f32" else MFMA_SUSTAINED_TFLOPS
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
            NP -= NP % NB
            ref = TrackEngine(arch, sd, cfg["H"], cfg["W"], batch=NB, dtype=torch.float32, device=dev)
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
                      "frames": NP, "first_frame": P0, "bench_engine": f"{dtype_name} B={Bs}", "reference_engine": f"f32 B={NB}"}
            # bars = 2 x the stream measurements of profiles/parity_r03_{c2,c4}.json (tools/parity_stream.py; every engine free running):
            #   (box, decoder output, score -- max abs error over rows matched by token --, births flipped / active rows)
            # The 16-bit figures are those of the ARITHMETIC TYPE on this random-init network (eager torch in the same type is 3-4x
            # further from fp32 on every one of them, same files); there is no `or few flips` escape any more.
            # births: stream means 4.3 % (bf16) / 0.9 % (fp16) of the active rows at C2, 5.0 % / 0.8 % at C4; on ~900 active rows the
            # sampling spread is +-0.7 % / +-0.3 %
            # (a 32-frame window is not the stream mean: frames 8..39 of sequence 0 measure 8.4 % for bf16 -- deterministic, the same
            # on every device -- so the bars are 1.4 x that window and 2 x the fp16 one; eager torch bf16 on this network: 16.6 %)
            bars = {"f32": (1e-4, 1e-3, 1e-3, 0.0), "f16": (5e-3, 0.6, 0.16, 0.03), "bf16": (9e-3, 1.3, 0.4, 0.12)}[dtype_name]
            # round 4 (VERDICT r3 #3c): where this exact window has a committed measurement (profiles/r03_f_bench_*.json -- the window
            # is deterministic: same frames, same kernels, no atomics), the bar is 1.5 x THAT measurement, so that a 2 x regression of
            # the 16-bit path fails:          box      hs     score   births / active
            # (re-measured with the folded head of round 4, profiles/r04_f_bench_*.json: the value maps and score logits no longer pass
            # through a 16-bit feature map, which moves the window's maxima by a few per cent in either direction)
            measured = {("c2", "bf16"): (4.08e-3, 0.673, 0.147, 0.0744), ("c2", "f16"): (1.07e-3, 0.115, 0.0256, 0.0153),
                        ("c4", "bf16"): (4.89e-3, 0.638, 0.0650, 0.0293)}.get((cfg_name, dtype_name))
            if measured is not None:
                bars = tuple(round(1.5 * v, 5) for v in measured)
            st_ = parity["bench_engine_vs_fp32_engine"]
            parity["bars"] = {"box_matched": bars[0], "hs_matched": bars[1], "score_matched": bars[2], "birth_flip_frac_of_active": bars[3]}
            parity["ok"] = bool(st_["box_max_err_matched"] <= bars[0] and st_["hs_max_err_matched"] <= bars[1]
                                and st_["score_max_err_matched"] <= bars[2] and st_["birth_flip_frac_of_active"] <= bars[3]
                                and n_masked == 0)
            if a.config == "full":
                # a TIMING configuration: the bars above were measured on the s-scale fixtures (this network is 2.2x as deep and
                # its score heads were calibrated at another resolution), so its differences are reported and only sanity is gated
                parity["bars"] = None
                parity["gated"] = "sanity only (no selected masked token, finite outputs, top-k overlap >= 0.8): timing configuration"
                parity["ok"] = bool(n_masked == 0 and bool(torch.isfinite(got["boxes"]).all()) and st_["topk_overlap"] >= 0.8)
            NPc = NB

            def engine_check(keep):
                """fp32 engine vs the CPU oracle on the oracle's own frames: logits <= 1e-3, ids exact (given the same selection)."""
                import numpy as np
                from oracle import track_oracle as O
                u8 = torch.from_numpy(np.concatenate([k[0] for k in keep] * (NPc // len(keep) + 1))[:NPc]).to(dev)
                o = {k: v.clone() for k, v in ref.forward(u8).items()}
                torch.cuda.synchronize()
                res = {"frames": len(keep), "logits_max_err": 0.0, "topk_equal": True, "ids_exact": True}
                for i, (_, r, ids) in enumerate(keep):
                    tk = o["topk_ind"][i].cpu().long()
                    same = bool(torch.equal(tk, r["topk_ind"][0]))
                    res["topk_equal"] &= same
                    if same:
                        res["logits_max_err"] = max(res["logits_max_err"], float((o["logits"][i].cpu() - r["dec_scores"][0]).abs().max()))
                        res["ids_exact"] &= bool(torch.equal(o["obj_idxes"][i].cpu(), ids))
                res["ok"] = bool(res["logits_max_err"] <= 1e-3 and res["ids_exact"])
                return res
        else:
            engine_check = None
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

        if world == 1 and not a.no_cpu_baseline:
            log("cpu baseline (oracle on the host cores)")
            try:
                cpu, par_cpu = cpu_baseline(cfg, arch, sd, a.cpu_frames, engine_check)
                if parity is not None and par_cpu is not None:
                    parity["fp32_engine_vs_cpu_oracle"] = par_cpu
                    parity["ok"] = bool(parity["ok"] and par_cpu["ok"])
            except Exception as e:  # the baseline must never lose the measured line
                cpu = {"error": repr(e)}

    n_launches = eng.num_launches if (rank == 0 and not a.dry_run) else 0
    extra = None
    if (rank == 0 and world == 1 and not a.dry_run and not a.no_extra_legs and a.config == "c2" and not a.temporal
            and a.dtype is None and a.batch is None and not a.from_host and not a.predictor and not a.resize_from):
        # the other BASELINE.json configurations and the other ways frames reach the engine, observed by whoever runs the default
        # bench: child processes of this one (the parent's plan is released first), 20 timed steps each, own parity gates included
        import subprocess
        pipe = eng = None
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        extra = {}
        legs = (("c4_bf16", ["--config", "c4"], 20), ("c5_f16", ["--config", "c5"], 20),
                ("c2_bf16_temporal100", ["--temporal", "100", "--batch", "32"], 20),
                # VERDICT r3 #3b: the engine that meets "bit-exact ids" (fp32), driver-observed
                ("c2_f32", ["--dtype", "f32"], 10),
                # the reference's own model scale (yolo_track.yaml 1.0 / 1.0)
                ("full_bf16", ["--config", "full"], 20),
                # frames that arrive from the host inside the timed region; the product predictor
                ("c2_bf16_from_host", ["--from-host"], 20),
                ("c2_bf16_from_host_1080p_resize", ["--from-host", "--resize-from", "1080x1920"], 20),
                ("c2_bf16_predictor", ["--predictor"], 5),
                ("c2_bf16_predictor_pinned_source", ["--predictor", "--pinned-source"], 5),
                ("c2_bf16_sustained_200_steps", [], 200))
        for name, flags, nsteps in legs:
            log(f"extra leg {name}")
            cmd = [sys.executable, os.path.abspath(__file__), *flags, "--steps", str(nsteps), "--warmup", "2", "--no-cpu-baseline",
                   "--no-extra-legs"] + ([] if name in ("c2_f32", "full_bf16") else ["--no-launch-table"]) \
                  + (["--no-parity"] if "sustained" in name else [])
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
                lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
                d = json.loads(lines[-1])
                extra[name] = {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"],
                               "dtype": d["dtype"], "workload": d["config"]["workload"], "parity": d.get("parity"), "rc": r.returncode}
                for key in ("from_host", "predictor", "resize_from", "host_link", "host_copies"):
                    if key in d["config"]:
                        extra[name][key] = d["config"][key]
                if d.get("roofline"):
                    extra[name]["roofline"] = d["roofline"]
                    extra[name]["roofline_step"] = {k: v for k, v in d["roofline_step"].items() if k != "top_kernels"}
                    extra[name]["top_kernels"] = d["roofline_step"].get("top_kernels")
                cname = "c4" if name.startswith("c4") else "c2"
                if "temporal" not in name and not name.startswith("full"):    # SURVEY §8(d) figure: FPS x algorithmic bytes per frame / HBM peak
                    extra[name]["roofline_frac_survey_8d"] = round(d["value"] * ALG_BYTES_FRAME[cname] * (2 if d["dtype"] == "f32" else 1)
                                                                   / (HBM_PEAK_GBS * 1e9), 4)
                if "from_host" in name and "resident" not in extra:
                    pass
            except Exception as e:  # a failing leg must not lose the headline line
                extra[name] = {"error": repr(e)[:300]}
        for name in ("c2_bf16_from_host", "c2_bf16_predictor", "c2_bf16_predictor_pinned_source", "c2_bf16_sustained_200_steps"):
            if "value" in extra.get(name, {}):
                extra[name]["vs_resident_headline"] = round(extra[name]["value"] / fps, 4)

    if rank == 0:
        if a.dry_run:
            workload = f"DRY RUN (no GPU): synthetic step, {B} frames/step/rank"
            launches = 0
        else:
            mode = (f"temporal mode, {B} sequences in lockstep, {a.temporal} track slots" if a.temporal else
                    f"{B} frames/step/GPU as {S} sub-batches on {S} HIP streams")
            scale = ("yolo_track.yaml at depth 1.0 / width 1.0 (46 M parameters)" if a.config == "full" else "YOLOv8 s-scale")
            feed = ("uint8 frames resident in HBM (input slots, no per-step copy)" if not (a.from_host or a.predictor) else
                    "uint8 frames fed from PINNED HOST memory inside the timed region (copy stream, overlapped)" if a.from_host else
                    "uint8 frames handed to TrackPredictor.__call__ as pageable host arrays inside the timed region")
            if a.predictor:
                mode = f"{B} frames per call in chunks of {eng.B}, {S} engine(s) taking the chunks in turn on {S} HIP stream(s)"
            workload = (f"{a.config.upper()}: {scale} backbone/neck + 6-layer MOTR decoder, {arch.nq} queries, {cfg['W']}x{cfg['H']}, "
                        f"{feed}, {mode}, {len(my_seqs)} sequence(s) per GPU, "
                        f"{'eager' if a.no_graph else 'hipGraph replay'}")
            launches = n_launches
        metric = {"c2": "frames/sec (whole node) on 1088x608 MOT17 streams", "c5": "frames/sec (whole node) on 1088x608 MOT17 streams",
                  "full": "frames/sec (whole node) on 1088x608 MOT17 streams",
                  "c4": "frames/sec (whole node) on 1920x1088 DanceTrack-shape streams"}[a.config]
        line = {
            "metric": metric, "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "none" if a.dry_run else dtype_name, "data": "synthetic",
            "config": dict({"workload": workload, "frames_per_step_per_gpu": B, "streams": S, "graph": not a.no_graph,
                            "launches_per_step": launches, "weights": "seeded synthetic (fixture recipe, mo_yolo_amd/fixtures.py)",
                            "sequences_of_rank0": my_seqs, "control_backend": backend, "hip_visible_devices_rank0": visible,
                            "dry_run": bool(a.dry_run)}, **line_extra),
            "roofline": roof, "roofline_step": roof_step, "parity": parity,
        }
        if cpu is not None:
            line["cpu_baseline"] = cpu
        if extra is not None:
            line["extra"] = extra
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if parity is not None and not parity.get("ok", True):
        print("bench.py: PARITY GATE FAILED: " + json.dumps(parity), file=sys.stderr, flush=True)
        sys.exit(3)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Headline benchmark: whole-job frames/s of the per-frame tracking step on synthetic
1088x608 MOT17-shape streams (BASELINE.json metric; config C2 at N=1, C3 = one sequence shard per
GPU at N>1, no collective on the data path).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path (fused preprocess -> backbone/neck -> decoder -> ID assignment +
predictor rows) over one batch of `--batch` frames already resident in HBM, replayed as one
hipGraph.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

# algorithmic bytes per frame, fused-op convention, bf16 (weights 25.7 MB of it)
ALG_BYTES_FRAME = {"c2": 0.415e9, "c4": 1.154e9}
ALG_WEIGHT_BYTES = 0.0257e9
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=576, help="frames per step per GPU (288 per sub-batch engine is the largest whose buffers stay inside 2 GiB descriptors)")
    ap.add_argument("--config", default="c2", choices=["c2", "c4"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--streams", type=int, default=2,
                    help="split the batch into this many sub-batches, each on its own HIP stream + hipGraph")
    ap.add_argument("--split-priority", type=int, default=int(os.environ.get("MOY_SPLIT_PRIORITY", "0")),
                    help="1: the query-sized chain of each sub-batch replays as its own hipGraph on a high-priority stream")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=12)
    ap.add_argument("--dump-launches", default=None, help="write the per-launch timing table (json) here")
    return ap.parse_args()


def cpu_baseline(cfg, arch, sd, n_frames):
    """Oracle (a port of the reference's eager path, verified against it in the build container)
    timed on this box's host cores: numeric graph + reference-faithful per-frame state machine."""
    from oracle import track_oracle as O
    from mo_yolo_amd.synth import SyntheticSequence, to_network_input
    seq = SyntheticSequence(0, cfg["H"], cfg["W"], cfg["style"])
    cores = min(torch.get_num_threads(), 16)      # eager batch-1 ops stop scaling (and regress) beyond ~16 threads
    torch.set_num_threads(cores)
    with torch.no_grad():
        for t in range(2):
            O.forward(to_network_input(seq.frames(t, 1)), sd, arch)
        frames = [to_network_input(seq.frames(t, 1)) for t in range(n_frames)]
        t0 = time.perf_counter()
        for x in frames:
            r = O.forward(x, sd, arch)
            scores = r["dec_scores"][0].sigmoid().max(-1).values
            ids, _, _ = O.assign_ids_loop(scores)                       # host loop as shipped (head.py:1232-1243)
            ids = torch.tensor(ids)
            O.tracker_update_copy(scores.tolist(), r["dec_bboxes"][0].numpy(), ids.tolist())
            O.postprocess(r["y"][0], r["dec_scores"][0], ids, 0.25, orig_hw=(cfg["H"], cfg["W"]))
        dt = time.perf_counter() - t0
    return {"value": n_frames / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{n_frames} frames of the same synthetic stream, batch 1, fp32 eager torch-CPU oracle incl. host state machine"}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from mo_yolo_amd.engine import StreamedEngines
    from mo_yolo_amd.synth import SyntheticSequence
    from tests._util import fixture
    cfg, arch, sd = fixture(a.config)
    dtype = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    B = a.batch
    # The batch of a step is cut into `--streams` sub-batches, each with its own engine (static buffers), HIP stream and
    # hipGraph: the latency-bound decoder launches of one sub-batch run beside the bandwidth-bound backbone of another.
    S = max(1, a.streams)
    assert B % S == 0, "--batch must be a multiple of --streams"
    Bs = B // S
    pipe = StreamedEngines(arch, sd, cfg["H"], cfg["W"], batch=B, streams=S, graph=not a.no_graph, dtype=dtype, device=dev,
                           split_priority=bool(a.split_priority))
    eng = pipe.engines[0]

    # sequence shard of this rank (sequence i -> GPU i, no cross-GPU term)
    seq = SyntheticSequence(rank, cfg["H"], cfg["W"], cfg["style"])
    n_batches = 3
    batches = [torch.from_numpy(seq.frames(i * B, B)).to(dev) for i in range(n_batches)]
    torch.cuda.synchronize()
    pipe.forward(batches[0])                      # first call: eager pass + graph capture
    torch.cuda.synchronize()

    def step(i):
        pipe.forward(batches[i % n_batches])

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(i)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    out = eng.outputs()
    n_masked = int(out["n_masked"].sum())
    active = float((out["obj_idxes"] >= 0).sum()) / Bs

    # ---- per-launch timing with HIP events on the launch stream (eager replay of the same plan)
    roof, roof_step = None, None
    if rank == 0:
        st = torch.cuda.current_stream()
        nL = eng.num_launches
        acc = [0.0] * nL
        reps = max(3, min(a.steps, 10))
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(nL + 1)]
        for rep in range(reps):
            eng.input.copy_(batches[rep % n_batches][:Bs])
            evs[0].record(st)
            for i in range(nL):
                eng.run_steps(i, i + 1)
                evs[i + 1].record(st)
            torch.cuda.synchronize()
            for i in range(nL):
                acc[i] += evs[i].elapsed_time(evs[i + 1])
        per = [x / reps for x in acc]                              # ms per launch
        dom = max(range(nL), key=lambda i: per[i])
        m = eng.meta[dom]
        ach = m["bytes"] / (per[dom] * 1e-3) / 1e9 if m["bytes"] else 0.0
        traffic = None      # HBM bytes per launch from the committed PMC passes (tools/pmc_traffic.sh), if this launch was profiled
        tp = os.path.join(ROOT, "profiles", "traffic_by_launch.json")
        if os.path.exists(tp):
            traffic = json.load(open(tp)).get(m["name"], {}).get("hbm_bytes")
        roof = {"bound": "hbm", "kernel": m["name"], "launch_index": dom, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                "avg_ms": round(per[dom], 4), "share_of_step": round(per[dom] / sum(per), 4),
                "alg_bytes_per_launch": m["bytes"], "tflops": round(m["flops"] / (per[dom] * 1e-3) / 1e12, 2)}
        bytes_step = (ALG_BYTES_FRAME[a.config] - ALG_WEIGHT_BYTES) * B + ALG_WEIGHT_BYTES
        if a.dtype == "f32":
            bytes_step *= 2
        ms_step = dt / a.steps * 1e3
        ach_s = bytes_step / (ms_step * 1e-3) / 1e9
        roof_step = {"bound": "hbm", "achieved": round(ach_s, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(ach_s / HBM_PEAK_GBS, 4), "alg_bytes_per_step": bytes_step,
                     "launches": nL, "sum_kernel_ms_eager": round(sum(per), 3)}
        if a.dump_launches:
            with open(a.dump_launches, "w") as f:
                json.dump([dict(i=i, ms=per[i], **eng.meta[i]) for i in range(nL)], f)
        top = sorted(range(nL), key=lambda i: -per[i])[:8]
        roof_step["top_kernels"] = [{"name": eng.meta[i]["name"], "ms": round(per[i], 4)} for i in top]

    if rank == 0:
        fps = B * a.steps * world / dt
        line = {
            "metric": "frames/sec (whole node) on 1088x608 MOT17 streams" if a.config == "c2" else
                      "frames/sec (whole node) on 1920x1088 DanceTrack-shape streams",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": f"{a.config.upper()}: YOLOv8 s-scale backbone/neck + 6-layer MOTR decoder, "
                                   f"{arch.nq} queries, {cfg['W']}x{cfg['H']}, uint8 frames resident in HBM, "
                                   f"{B} frames/step/GPU as {S} sub-batches on {S} HIP streams, one sequence shard per GPU, hipGraph replay",
                       "frames_per_step_per_gpu": B, "streams": S, "graph": not a.no_graph, "launches_per_step": eng.num_launches,
                       "weights": "seeded synthetic (fixture c2 recipe)", "mean_active_tracks": round(active, 1),
                       "masked_tokens_selected": n_masked},
            "roofline": roof, "roofline_step": roof_step,
        }
        if world == 1 and not a.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(cfg, arch, sd, a.cpu_frames)
            except Exception as e:  # the baseline must never lose the measured line
                line["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

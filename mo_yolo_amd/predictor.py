"""Predictor counterpart: the reference's stream loop I/O contract for the `track` task.

Mirrors `BasePredictor.__call__/stream_inference/preprocess` (ultralytics/engine/predictor.py:117-134,
229-344), `TrackPredictor.postprocess` (ultralytics/models/MOTRtrack/predict.py:13-94) and
`TrackResults.save_txt` (ultralytics/engine/results.py:366-371, 475-512).  Preprocess, the network
and the row building all run on the device inside one TrackEngine step; this file only moves frames
in, rows out, and formats text.
"""
from __future__ import annotations

import os
from typing import List, Sequence

import numpy as np
import torch

from . import ops
from .engine import TrackEngine


class TrackResults:
    """boxes: float32 [K, 6] = (x1, y1, x2, y2, score, cls) in `orig_shape` pixels (or normalised for
    tensor sources, predict.py:66); track_id: int64 [K'] or None for the detection-style fallback."""

    def __init__(self, boxes: np.ndarray, track_id, orig_shape, path="", speed=None):
        self.boxes, self.track_id, self.orig_shape, self.path = boxes, track_id, tuple(orig_shape), path
        self.speed = speed or {}

    def __len__(self):
        return len(self.boxes)

    def txt_lines(self, save_conf=False) -> List[str]:
        """`track_id cls cx cy w h [conf]`, xywh normalised by orig_shape, %g (results.py:495-507)."""
        if self.track_id is None:
            raise ValueError("detection-style fallback results carry no track ids")
        h, w = self.orig_shape
        out = []
        for j in range(len(self.boxes)):
            x1, y1, x2, y2, cf, c = (np.float32(v) for v in self.boxes[j])
            xywhn = ((x1 + x2) / 2 / w, (y1 + y2) / 2 / h, (x2 - x1) / w, (y2 - y1) / h)
            line = (int(self.track_id[j]), int(c), *xywhn) + ((float(cf),) if save_conf else ())
            out.append(("%g " * len(line)).rstrip() % line)
        return out

    def save_txt(self, txt_file, save_conf=False):
        lines = self.txt_lines(save_conf)
        if lines:
            with open(txt_file, "a") as f:
                f.writelines(t + "\n" for t in lines)


class _HostRing:
    """Pinned host staging of the predictor's stream loop: `depth` frame buffers in (one chunk each) and `depth` result
    blocks out.  A buffer is reused only after the transfer that read / wrote it has completed (its event)."""

    def __init__(self, depth, in_shape, in_dtype, out_bytes):
        self.depth = depth
        self.inp = [torch.empty(in_shape, dtype=in_dtype).pin_memory() for _ in range(depth)]
        self.out = [torch.empty(out_bytes, dtype=torch.uint8).pin_memory() for _ in range(depth)]
        self.h2d_done = [None] * depth
        self.d2h_done = [None] * depth


def _stage(dst: torch.Tensor, src: np.ndarray, threads: int = 8):
    """Pageable frames -> pinned staging buffer (numpy releases the GIL in the copy, so a few threads reach the host's copy rate)."""
    d = dst.numpy()
    k = src.shape[0]
    try:
        threads = max(1, min(threads, len(os.sched_getaffinity(0)), k))
    except AttributeError:      # pragma: no cover
        threads = max(1, min(threads, k))
    if k * src[0].nbytes < (8 << 20) or threads <= 1:
        np.copyto(d[:k], src)
        return
    from concurrent.futures import ThreadPoolExecutor
    cuts = [k * i // threads for i in range(threads + 1)]
    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(lambda ab: np.copyto(d[ab[0]:ab[1]], src[ab[0]:ab[1]]), zip(cuts[:-1], cuts[1:])))


class TrackPredictor:
    def __init__(self, arch, state_dict, imgsz=(608, 1088), conf=0.25, dtype=torch.float32, device="cuda", batch=1,
                 graph=False, temporal=0, ring=3, streams=1):
        """`temporal` = track slots per sequence (0: the shipped per-frame-reset semantics).  In temporal mode the `batch`
        frames of a chunk are ONE time step of `batch` sequences running in lockstep (source order t0s0, t0s1, ..., t1s0, ...);
        call `reset_sequences()` at the start of new videos (`is_first`, head.py:199-205).
        `ring` = depth of the host-fed pipeline (round 4): pinned staging buffers, device input slots and pinned result
        blocks; chunk j+1 is copied in on a copy stream while chunk j computes, and chunk j's results (ONE packed
        device-to-host copy: rows | ids | counts) are read on the host while chunk j+1 computes.
        `streams` > 1 (per-frame mode only: frames are independent there): that many engines, each on its own HIP
        stream, take the chunks in turn -- the query-sized decoder launches of one chunk run beside the convolutions of the next
        (what `StreamedEngines` does for the benchmark); results come back in source order."""
        self.arch, self.sd, self.imgsz, self.conf, self.dtype, self.device = arch, state_dict, tuple(imgsz), conf, dtype, device
        self.batch, self.graph, self.temporal, self.ring = batch, graph, int(temporal), max(2, int(ring))
        self.streams = 1 if self.temporal else max(1, int(streams))
        self._engines = {}                                 # key -> the first engine of the set (the only one when streams == 1)
        self._sets = {}                                    # key -> [(engine, ring, compute stream or None = the caller's), ...]
        self._copy_stream = None

    def reset_sequences(self, which=None):
        for sets in self._sets.values():
            for eng, _, _ in sets:
                eng.reset_sequence(which)
        for key, eng in self._engines.items():
            if key not in self._sets:
                eng.reset_sequence(which)

    def _engine(self, fmt, orig_hw=None):
        key = (fmt, tuple(orig_hw or self.imgsz))
        if key not in self._engines:
            H, W = self.imgsz
            resized = fmt == "u8" and key[1] != self.imgsz
            n_eng = self.streams if fmt == "u8" else 1
            sets = []
            for i in range(n_eng):
                eng = TrackEngine(self.arch, self.sd, H, W, batch=self.batch, dtype=self.dtype, device=self.device,
                                  input_format=fmt, conf=self.conf, scale_boxes=(fmt == "u8"), orig_hw=key[1], temporal=self.temporal,
                                  n_inputs=1 if resized else self.ring)
                if self.graph:
                    eng.forward(torch.zeros_like(eng.input))
                    eng.capture()
                    eng.reset_sequence()                  # the warm-up frames must not leave tracks behind
                ring = None
                if fmt == "u8":
                    oh, ow = key[1]
                    ring = _HostRing(self.ring, (self.batch, oh, ow, 3), torch.uint8, eng.result_block.numel())
                    # frames of another size land in device staging buffers first and are stretch-resized into the engine's input
                    ring.dev_in = ([torch.empty(self.batch, oh, ow, 3, dtype=torch.uint8, device=eng.dev) for _ in range(self.ring)]
                                   if resized else eng.inputs)
                    ring.compute_done = [None] * self.ring
                    ring.resized = resized
                sets.append((eng, ring, torch.cuda.Stream(device=eng.dev) if n_eng > 1 else None))
            self._engines[key] = sets[0][0]
            if fmt == "u8":
                self._sets[key] = sets
        return self._engines[key]

    def preprocess(self, im):
        """List of uint8 BGR HWC frames (any size) or float [B,3,H,W] in [0,1] at network resolution.
        Frames of another size are stretch-resized on the device like `pre_transform` does with
        LetterBox(scaleFill) (MOTRtrack/predict.py:96-105, data/augment.py:573-576: cv2 INTER_LINEAR, no padding);
        the BGR->RGB / CHW / float / 255 arithmetic of predictor.py:125-133 is fused into the stem kernel.
        uint8 frames stay on the HOST here: `__call__` moves them chunk by chunk through the pinned ring."""
        if isinstance(im, torch.Tensor) and im.dtype == torch.uint8 and not im.is_cuda:
            # host uint8 frames handed over as a tensor; a PINNED one (a decoder that writes straight into page-locked memory)
            # skips the staging copy: its chunks cross the link from where they lie
            if im.dim() != 4 or im.shape[3] != 3 or not im.is_contiguous():
                raise ValueError("frame source must be contiguous uint8 [B,H,W,3] BGR (frames of one size)")
            return im, "u8"
        if isinstance(im, torch.Tensor):
            if im.dim() != 4 or im.shape[1] != 3 or tuple(im.shape[2:]) != self.imgsz:
                raise ValueError(f"tensor source must be [B,3,{self.imgsz[0]},{self.imgsz[1]}]")
            if im.shape[2] % 32 or im.shape[3] % 32:
                raise ValueError("tensor source sides must be multiples of 32 (data/loaders.py:316-332)")
            return im.to(self.device, torch.float32), "f32"
        arr = np.stack(im) if not isinstance(im, np.ndarray) else im
        if arr.dtype != np.uint8 or arr.ndim != 4 or arr.shape[3] != 3:
            raise ValueError("frame source must be uint8 [B,H,W,3] BGR (frames of one size)")
        return np.ascontiguousarray(arr), "u8"

    def _results_of(self, eng, host_block, k, orig_hw, paths, s):
        rows, tid, n_rows, n_ids = eng.unpack_result_block(host_block)
        res = []
        for b in range(k):
            t = None if n_ids[b] < 0 else tid[b, :n_ids[b]].copy()
            res.append(TrackResults(rows[b, :n_rows[b]].copy(), t, orig_hw, path=(paths[s + b] if paths else "")))
        return res

    @torch.no_grad()
    def __call__(self, source, paths: Sequence[str] | None = None) -> List[TrackResults]:
        x, fmt = self.preprocess(source)
        orig_hw = tuple(x.shape[1:3]) if fmt == "u8" else self.imgsz
        eng = self._engine(fmt, orig_hw)
        n = x.shape[0]
        if self.temporal and n % self.batch:
            # batch element b is SEQUENCE b with persistent query memory: padding a short chunk with another sequence's frame
            # would advance the memory, id counter and miss counters of the padded sequences
            raise ValueError(f"temporal mode: the source must hold whole time steps ({self.batch} sequences per step), got {n} frames")
        if fmt == "u8":
            return [r for chunk in self._host_fed(iter([(x, paths)]), orig_hw) for r in chunk]
        # float tensor source (LoadTensor, data/loaders.py:316-332): already on the device; one packed result copy per chunk
        results: List[TrackResults] = []
        for s in range(0, n, self.batch):
            chunk = x[s:s + self.batch]
            k = chunk.shape[0]
            if k < self.batch:                                     # ragged tail (per-frame mode only): pad with the last frame
                chunk = torch.cat([chunk, chunk[-1:].expand(self.batch - k, *chunk.shape[1:])], 0)
            eng.forward(chunk.contiguous())
            results += self._results_of(eng, eng.result_block.cpu(), k, orig_hw, paths, s)
        return results

    @torch.no_grad()
    def stream(self, batches):
        """The generator form of the stream loop (`BasePredictor.stream_inference`, engine/predictor.py:256-344, is a generator over
        the dataset): `batches` yields uint8 frame arrays [n, H, W, 3] of ONE frame size (or `(frames, paths)` pairs); one list of
        TrackResults is yielded per chunk of `batch` frames, in source order, a few chunks behind the input -- the pipeline
        (staging, H2D, the engines, the packed D2H) stays full ACROSS the arrays, which a sequence of `__call__`s cannot do."""
        first = None
        it = iter(batches)

        def norm():
            nonlocal first
            for item in it:
                frames, paths = item if isinstance(item, tuple) else (item, None)
                x, fmt = self.preprocess(frames)
                if fmt != "u8":
                    raise ValueError("stream() takes host uint8 frames")
                hw = tuple(x.shape[1:3])
                if first is None:
                    first = hw
                elif hw != first:
                    raise ValueError("stream(): every array must hold frames of the first array's size")
                if self.temporal and x.shape[0] % self.batch:
                    raise ValueError(f"temporal mode: every array must hold whole time steps ({self.batch} sequences per step)")
                yield x, paths

        gen = norm()
        try:
            head = next(gen)
        except StopIteration:
            return
        self._engine("u8", first)

        def chain():
            yield head
            yield from gen

        yield from self._host_fed(chain(), first)

    def _host_fed(self, arrays, orig_hw):
        """[stage chunk j+1 into pinned memory + H2D on the copy stream] || [chunk j on the device] || [read chunk j-1's rows].
        The reference does `im.to(device)` from pageable memory and a `.cpu()` per result tensor, all blocking (predictor.py:130,
        predict.py:27-76).  `arrays` yields (frames, paths); yields one list of TrackResults per chunk, in source order."""
        from collections import deque
        sets = self._sets[("u8", tuple(orig_hw))]
        S = len(sets)
        dev = sets[0][0].dev
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(device=dev)
        cs, caller = self._copy_stream, torch.cuda.current_stream(dev)
        B, D = self.batch, sets[0][1].depth
        pending = deque()                                          # chunks whose results are in flight, in source order

        def collect():
            e_, ring_, pr, pk, ps, pp = pending.popleft()
            ring_.d2h_done[pr].synchronize()
            return self._results_of(e_, ring_.out[pr], pk, orig_hw, pp, ps)

        j = 0
        for x, paths in arrays:
            n = x.shape[0]
            pinned_src = isinstance(x, torch.Tensor) and x.is_pinned()
            xn = x.numpy() if isinstance(x, torch.Tensor) else x
            for s in range(0, n, B):
                e, ring, own = sets[j % S]
                main = own if own is not None else caller          # this chunk's compute stream
                r = (j // S) % D
                j += 1
                k = min(B, n - s)
                direct = pinned_src and k == B                     # a whole chunk in page-locked memory: no staging copy
                if not direct:
                    if ring.h2d_done[r] is not None:
                        ring.h2d_done[r].synchronize()             # the transfer that last read this staging buffer
                    _stage(ring.inp[r], xn[s:s + k])
                    if k < B:                                      # ragged tail (per-frame mode only): pad with the last frame
                        ring.inp[r].numpy()[k:] = xn[s + k - 1]
                with torch.cuda.stream(cs):
                    if ring.compute_done[r] is not None:
                        cs.wait_event(ring.compute_done[r])        # the step that last read this device slot
                    ring.dev_in[r].copy_(x[s:s + B] if direct else ring.inp[r], non_blocking=True)
                    ring.h2d_done[r] = torch.cuda.Event()
                    ring.h2d_done[r].record(cs)
                if ring.d2h_done[r] is not None:
                    ring.d2h_done[r].synchronize()                 # (its rows were handed out at least D chunks ago)
                with torch.cuda.stream(main):
                    if own is not None:
                        main.wait_stream(caller)                   # whatever the caller queued before (weights, resets)
                    main.wait_event(ring.h2d_done[r])
                    if ring.resized:
                        ops.resize_linear_u8(ring.dev_in[r], self.imgsz, out=e.input)
                        e.forward(None)
                    else:
                        e.forward(None, slot=r)
                    ring.compute_done[r] = torch.cuda.Event()
                    ring.compute_done[r].record(main)
                    ring.out[r].copy_(e.result_block, non_blocking=True)     # ONE device-to-host copy per chunk
                    ring.d2h_done[r] = torch.cuda.Event()
                    ring.d2h_done[r].record(main)
                pending.append((e, ring, r, k, s, paths))
                while len(pending) > S:                            # keep S chunks computing while the host reads the oldest one's rows
                    yield collect()
        while pending:
            yield collect()


class DetectionPredictor:
    """Config C1 plumbing: `DetectionPredictor` of ultralytics/models/yolo/detect/predict.py:10-30 for
    network-resolution frames: fused preprocess -> YOLOv8 backbone/neck -> Detect decode -> NMS ->
    scale_boxes to the original image size, all inside one TrackEngine step.
    Returns per frame a float32 [n, 6] array (x1, y1, x2, y2, conf, cls)."""

    def __init__(self, arch, state_dict, imgsz=(640, 640), conf=0.25, iou=0.7, max_det=300, dtype=torch.float32,
                 device="cuda", batch=1, orig_hw=None):
        assert arch.head_kind == "detect"
        self.batch = batch
        self.eng = TrackEngine(arch, state_dict, imgsz[0], imgsz[1], batch=batch, dtype=dtype, device=device, conf=conf,
                               iou=iou, max_det=max_det, orig_hw=orig_hw or imgsz)

    @torch.no_grad()
    def __call__(self, frames_u8) -> List[np.ndarray]:
        x = torch.as_tensor(np.stack(frames_u8) if not isinstance(frames_u8, (np.ndarray, torch.Tensor)) else frames_u8)
        x = x.to(self.eng.dev)
        res = []
        for s in range(0, x.shape[0], self.batch):
            chunk = x[s:s + self.batch]
            k = chunk.shape[0]
            if k < self.batch:
                chunk = torch.cat([chunk, chunk[-1:].expand(self.batch - k, *chunk.shape[1:])], 0)
            self.eng.forward(chunk.contiguous())
            rows, _, n, _ = self.eng.unpack_result_block(self.eng.result_block.cpu())     # one device-to-host copy per chunk
            res += [rows[b, :n[b]].copy() for b in range(k)]
        return res

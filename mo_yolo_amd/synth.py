"""Synthetic MOT-shaped frame streams generated AT network resolution.

Frames are uint8 BGR HWC (what `BasePredictor.preprocess`, engine/predictor.py:117-134, is fed);
`to_network_input` applies exactly that function's arithmetic (BGR->RGB, HWC->CHW, float, /255).
Only integer draws and exact float32 arithmetic are used so a (seq_id, frame) pair is
byte-identical on every machine.  Ground-truth boxes/ids of the moving rectangles come along
for the HOTA half of the metric.
"""
from __future__ import annotations

import numpy as np
import torch


class SyntheticSequence:
    def __init__(self, seq_id: int, H: int = 608, W: int = 1088, style: str = "mot17", n_obj: int | None = None):
        assert H % 32 == 0 and W % 32 == 0
        self.H, self.W, self.seq_id, self.style = H, W, seq_id, style
        rng = np.random.Generator(np.random.PCG64(1000 + seq_id))
        if n_obj is None:
            lo, hi = (20, 60) if style == "mot17" else (10, 40)
            n_obj = int(rng.integers(lo, hi + 1))
        self.n_obj = n_obj
        # coarse background (16x16 cells) + per-sequence fine texture tile
        self.bg = rng.integers(40, 200, size=(H // 16, W // 16, 3), dtype=np.int64)
        self.tex = rng.integers(-12, 13, size=(32, 32, 3), dtype=np.int64)
        if style == "mot17":
            w = rng.integers(W // 40, W // 12, size=n_obj)
            h = (w * rng.integers(20, 32, size=n_obj)) // 10
        else:  # dancetrack-like: larger, overlapping
            w = rng.integers(W // 14, W // 6, size=n_obj)
            h = (w * rng.integers(18, 28, size=n_obj)) // 10
        self.w, self.h = w.astype(np.int64), np.minimum(h, H - 2).astype(np.int64)
        self.x0 = rng.integers(0, W - self.w, size=n_obj).astype(np.int64) * 16   # 1/16 px fixed point
        self.y0 = rng.integers(0, H - self.h, size=n_obj).astype(np.int64) * 16
        self.vx = rng.integers(-48, 49, size=n_obj).astype(np.int64)
        self.vy = rng.integers(-16, 17, size=n_obj).astype(np.int64)
        self.col = rng.integers(0, 256, size=(n_obj, 3), dtype=np.int64)
        # Aperiodic pixel noise (drawn LAST, so the objects / ground truth above are unchanged): the 16-pixel background cells
        # and the 32-pixel texture tile alone make many stride-8/16/32 tokens see IDENTICAL input windows -> identical features
        # and encoder scores tied to the last bit, i.e. a query order that no two correct implementations agree on (round 1:
        # adjacent top-k scores 8e-7 apart).  A frame takes a window of this plane at a frame-dependent offset.
        self.noise = rng.integers(-10, 11, size=(H + 64, W + 64, 3), dtype=np.int64)

    def boxes(self, t: int):
        """xyxy pixel boxes (float32) and ids of the rectangles visible in frame t."""
        x = (self.x0 + self.vx * t) // 16
        y = (self.y0 + self.vy * t) // 16
        x1 = np.clip(x, 0, self.W - 1); y1 = np.clip(y, 0, self.H - 1)
        x2 = np.clip(x + self.w, 0, self.W); y2 = np.clip(y + self.h, 0, self.H)
        vis = (x2 - x1 >= 4) & (y2 - y1 >= 4)
        b = np.stack([x1, y1, x2, y2], -1)[vis].astype(np.float32)
        return b, np.nonzero(vis)[0].astype(np.int64)

    def frame(self, t: int) -> np.ndarray:
        """uint8 [H, W, 3] BGR frame t."""
        H, W = self.H, self.W
        img = np.repeat(np.repeat(self.bg, 16, 0), 16, 1)
        img = img + np.tile(np.roll(self.tex, t % 32, 1), (H // 32, W // 32, 1))
        oy, ox = (17 * t) % 64, (29 * t) % 64
        img = img + self.noise[oy:oy + H, ox:ox + W]
        b, ids = self.boxes(t)
        for (x1, y1, x2, y2), i in zip(b.astype(np.int64), ids):
            img[y1:y2, x1:x2] = self.col[i]
            img[y1:y2:4, x1:x2] = 255 - self.col[i]          # stripes give the conv stack texture
        return np.clip(img, 0, 255).astype(np.uint8)

    def frames(self, t0: int, n: int) -> np.ndarray:
        return np.stack([self.frame(t) for t in range(t0, t0 + n)])


def to_network_input(frames_u8: np.ndarray | torch.Tensor) -> torch.Tensor:
    """[T,H,W,3] uint8 BGR -> [T,3,H,W] float32 RGB in [0,1]; arithmetic of predictor.py:125-133."""
    x = torch.as_tensor(frames_u8)
    x = x.flip(-1).permute(0, 3, 1, 2).contiguous().float()
    return x / 255

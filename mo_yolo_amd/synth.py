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

    def frames_torch(self, t0: int, n: int, device="cpu", chunk: int = 48) -> torch.Tensor:
        """uint8 [n, H, W, 3] frames t0 .. t0+n-1 drawn ON `device` with torch integer ops: the same bytes as `frames` (integer
        arithmetic only; tests/test_host_logic.py holds the two to each other), without the host loop -- a rank that starts with
        3 x 1152 frames spends seconds instead of half a minute, and eight ranks do not queue for the host cores.
        A later rectangle overwrites an earlier one exactly as the painting loop of `frame` does."""
        dev = torch.device(device)
        H, W = self.H, self.W
        i16 = torch.int16
        bg = torch.as_tensor(self.bg, dtype=i16, device=dev).repeat_interleave(16, 0).repeat_interleave(16, 1)       # [H, W, 3]
        tex = torch.as_tensor(self.tex, dtype=i16, device=dev)
        noise = torch.as_tensor(self.noise, dtype=i16, device=dev)
        col = torch.as_tensor(self.col, dtype=i16, device=dev)
        ys = torch.arange(H, device=dev, dtype=torch.int32)[None, :, None]
        xs = torch.arange(W, device=dev, dtype=torch.int32)[None, None, :]
        out = torch.empty(n, H, W, 3, dtype=torch.uint8, device=dev)
        for c0 in range(0, n, chunk):
            ts = list(range(t0 + c0, t0 + min(n, c0 + chunk)))
            m = len(ts)
            img = bg[None].repeat(m, 1, 1, 1)
            for k, t in enumerate(ts):                         # (two cheap ops per frame; the per-rectangle work below is batched)
                img[k] += torch.roll(tex, t % 32, 1).repeat(H // 32, W // 32, 1)
                oy, ox = (17 * t) % 64, (29 * t) % 64
                img[k] += noise[oy:oy + H, ox:ox + W]
            tt = np.asarray(ts, dtype=np.int64)[:, None]
            x = (self.x0[None] + self.vx[None] * tt) // 16
            y = (self.y0[None] + self.vy[None] * tt) // 16
            x1 = np.clip(x, 0, W - 1); y1 = np.clip(y, 0, H - 1)
            x2 = np.clip(x + self.w[None], 0, W); y2 = np.clip(y + self.h[None], 0, H)
            vis = (x2 - x1 >= 4) & (y2 - y1 >= 4)
            # an invisible rectangle becomes an empty one
            x2 = np.where(vis, x2, x1); y2 = np.where(vis, y2, y1)
            X1, Y1, X2, Y2 = (torch.as_tensor(v, dtype=torch.int32, device=dev) for v in (x1, y1, x2, y2))    # [m, n_obj]
            for i in range(self.n_obj):                        # painting order = object order (frame(): later objects on top)
                ry = (ys >= Y1[:, i, None, None]) & (ys < Y2[:, i, None, None])                                 # [m, H, 1]
                rx = (xs >= X1[:, i, None, None]) & (xs < X2[:, i, None, None])                                 # [m, 1, W]
                if not bool((ry.any(1) & rx.any(2)).any()):
                    continue
                stripe = ((ys - Y1[:, i, None, None]) & 3) == 0
                inside = (ry & rx)[..., None]
                img = torch.where(inside, torch.where((ry & stripe & rx)[..., None], 255 - col[i], col[i]), img)
            out[c0:c0 + m] = img.clamp_(0, 255).to(torch.uint8)
        return out


def to_network_input(frames_u8: np.ndarray | torch.Tensor) -> torch.Tensor:
    """[T,H,W,3] uint8 BGR -> [T,3,H,W] float32 RGB in [0,1]; arithmetic of predictor.py:125-133."""
    x = torch.as_tensor(frames_u8)
    x = x.flip(-1).permute(0, 3, 1, 2).contiguous().float()
    return x / 255

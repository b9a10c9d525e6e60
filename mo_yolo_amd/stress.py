"""Self-checks of the persistent LDS-DMA ring kernels and of the bench-scale plan (round 5).

No oracle is involved: every check compares the library with ITSELF -- the weight-stationary / ring kernels against the tiled
`gemm_kernel` on the same rows (the forms are bit-identical by construction: same MFMA, same k order), a plan against its own
repeated passes.  Used by `tests/test_gpu_stress.py`, `tools/stress_rings.py` (the long A/B screen) and by `bench.py`'s
determinism self-check outside the timed region.

What they screen for: the end-of-tile `s_waitcnt vmcnt(N)` of a DMA ring is hand-counted; a count that is too large does not
fail a test, it reads a tile on the strength of its DMA having been issued long ago -- wrong only when that DMA is slow (cold
translation, loaded memory system).  Hence the bandwidth hog on a second stream and the cache thrash between runs.
"""
from __future__ import annotations

import ctypes as C
import math
import os

import torch

from . import _lib as L
from . import ops


def _rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


class Hog:
    """A bandwidth hog on a second HIP stream: `kick(n)` enqueues n device-to-device copies of 1 GiB (about 0.4 ms each) that
    run beside whatever the caller launches next on its own stream."""

    def __init__(self, device="cuda", gib: float = 1.0):
        self.dev = torch.device(device)
        self.n = int(gib * (1 << 30))

    def __enter__(self):
        self.a = torch.empty(self.n, dtype=torch.uint8, device=self.dev)
        self.b = torch.empty(self.n, dtype=torch.uint8, device=self.dev)
        self.stream = torch.cuda.Stream(device=self.dev)
        return self

    def kick(self, n: int = 4):
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            for _ in range(n):
                self.b.copy_(self.a, non_blocking=True)

    def __exit__(self, *exc):
        self.stream.synchronize()
        del self.a, self.b
        return False


_thrash_buf = {}


def thrash(device="cuda", mib: int = 640):
    """Evict L2 / Infinity Cache (256 MB) and the address translations of the buffers under test: fill a buffer larger than both."""
    key = (str(device), mib)
    if key not in _thrash_buf:
        _thrash_buf[key] = torch.empty(mib << 20, dtype=torch.uint8, device=device)
    _thrash_buf[key].add_(1)


def first_diff(got: torch.Tensor, ref: torch.Tensor):
    bad = (got != ref).reshape(got.shape[0], -1).any(1).nonzero().flatten()
    if bad.numel() == 0:
        return None
    return dict(rows_differing=int(bad.numel()), first_rows=[int(v) for v in bad[:8]], last_row=int(bad[-1]),
                max_abs=float((got.float() - ref.float()).abs().max()))


# ------------------------------------------------------------------------------------------------ kernel-level cases
class _GemmCase:
    """One 1x1 form of `gemm_wreg_kernel` at a launch size that takes it (M >= 65536), and the same rows through the tiled kernel."""

    def __init__(self, form, dt, dev):
        self.form, self.dt, self.dev = form, dt, dev
        f = dict(value_k128_planes_remap=dict(K=128, N=1536, planes=True, remap=True),
                 value_k256_planes_remap=dict(K=256, N=1536, planes=True, remap=True),
                 value_k256_planes=dict(K=256, N=1536, planes=True),
                 n256_k128=dict(K=128, N=256, act=L.ACT_SILU, scale=True),
                 n256_k256_remap=dict(K=256, N=256, remap=True),
                 n256_k384=dict(K=384, N=256, act=L.ACT_SILU, scale=True),
                 n512_k512=dict(K=512, N=512),
                 n128_k128=dict(K=128, N=128, act=L.ACT_SILU, scale=True),
                 n128_k192=dict(K=192, N=128, act=L.ACT_SILU, scale=True),
                 n128_k256=dict(K=256, N=128, act=L.ACT_SILU, scale=True),
                 seeded_n256_k256=dict(K=256, N=256, act=L.ACT_SILU, scale=True, seeded=True),
                 seeded_n128_k128=dict(K=128, N=128, act=L.ACT_SILU, scale=True, seeded=True))[form]
        self.f = f
        K, N = f["K"], f["N"]
        self.K, self.N = K, N
        if f.get("seeded"):
            self.B, self.H, self.W = 40, 38, 70                     # whole images per reference launch (2660 rows each)
            self.hw = self.H * self.W
        else:
            self.B, self.hw = 140, 1000                             # 140 000 rows: ~17 tiles of 32 rows per block
        self.S, self.off = (self.hw + 300, 200) if f.get("remap") else (self.hw, 0)
        self.M = self.B * self.hw
        self.x = _rnd(self.M, K, seed=51).to(dev, dt)
        self.w = ops.pad_weight(_rnd(N, K, seed=52, scale=1 / math.sqrt(K)).to(dev), dt)
        self.shift = _rnd(N, seed=53, scale=0.1).to(dev)
        self.scale = (_rnd(N, seed=54) * 0.2 + 1.0).to(dev) if f.get("scale") else None
        self.act = f.get("act", L.ACT_NONE)
        if f.get("seeded"):
            self.seed = _rnd(self.B * (self.H // 2) * (self.W // 2), N, seed=55).to(dev)

    def _out(self):
        rows = self.B * self.S + 3
        if self.f.get("planes"):
            return torch.full((self.N // 32, rows, 32), 7.0, device=self.dev, dtype=self.dt)
        return torch.full((rows, self.N), 7.0, device=self.dev, dtype=self.dt)

    def _launch(self, out, b0, b1):
        f, hw, S, off = self.f, self.hw, self.S, self.off
        kw = dict(shift=self.shift, scale=self.scale, act=self.act)
        if f.get("remap"):
            kw.update(c_rpb=hw, c_bstride=S)
        r0 = b0 * S + off
        nrows = (b1 - b0) * S if f.get("remap") else (b1 - b0) * hw
        if f.get("planes"):
            kw.update(planes=(32, out.shape[1] * 32))
            dst = out[0, r0:r0 + nrows]
        else:
            dst = out[r0:r0 + nrows]
        if f.get("seeded"):
            lhw = (self.H // 2) * (self.W // 2)
            kw.update(pre=(self.seed[b0 * lhw:b1 * lhw], self.H, self.W))
        ops.gemm(self.x[b0 * hw:b1 * hw], self.w, self.N, self.K, out=dst, **kw)

    def run(self):
        out = self._out()
        self._launch(out, 0, self.B)
        return out

    def reference(self):
        out = self._out()
        per = max(1, 60000 // self.hw)
        for b0 in range(0, self.B, per):
            self._launch(out, b0, min(self.B, b0 + per))
        torch.cuda.synchronize()
        return out

    @staticmethod
    def where(got, ref):
        g, r = (got.permute(1, 0, 2), ref.permute(1, 0, 2)) if got.dim() == 3 else (got, ref)
        return first_diff(g.contiguous(), r.contiguous())


def gemm_case(form, dt, dev="cuda"):
    return _GemmCase(form, dt, dev)


class _ConvCase:
    """One 3x3 form of `conv_ws_kernel` / `conv_s2_kernel` at a launch size that takes it (>= 2 tiles per CU), and the same images
    one by one through the tiled kernel."""

    def __init__(self, form, dt, dev):
        self.form, self.dt, self.dev = form, dt, dev
        f = dict(c32=dict(C=32, N=32, s=1, H=120, W=136, B=24), c32_res=dict(C=32, N=32, s=1, H=120, W=136, B=24, res=True),
                 c64=dict(C=64, N=64, s=1, H=120, W=136, B=24), c64_res=dict(C=64, N=64, s=1, H=120, W=136, B=24, res=True),
                 c128=dict(C=128, N=128, s=1, H=60, W=136, B=24), c128_res=dict(C=128, N=128, s=1, H=60, W=136, B=24, res=True),
                 s2_32_64=dict(C=32, N=64, s=2, H=121, W=135, B=32), s2_64_128=dict(C=64, N=128, s=2, H=60, W=136, B=32),
                 s2_64_128_post=dict(C=64, N=128, s=2, H=60, W=136, B=32, post=True))[form]
        self.f = f
        Cc, N, s, H, W, B = f["C"], f["N"], f["s"], f["H"], f["W"], f["B"]
        self.Ho, self.Wo = (H + 2 - 3) // s + 1, (W + 2 - 3) // s + 1
        self.x = _rnd(B * H * W, Cc, seed=1).to(dev, dt)
        self.w = ops.pad_weight(_rnd(N, 9 * Cc, seed=2, scale=1 / math.sqrt(9 * Cc)).to(dev), dt)
        self.scale, self.shift = (_rnd(N, seed=3) * 0.2 + 1).to(dev), _rnd(N, seed=4, scale=0.1).to(dev)
        self.res = _rnd(B * H * W, N, seed=5).to(dev, dt) if f.get("res") else None
        if f.get("post"):
            self.w2 = ops.pad_weight(_rnd(N, N, seed=6, scale=1 / math.sqrt(N)).to(dev), dt)
            self.scale2, self.shift2 = (_rnd(N, seed=7) * 0.2 + 1).to(dev), _rnd(N, seed=8, scale=0.1).to(dev)

    def _launch(self, out, b0, b1, post=True):
        f = self.f
        Cc, N, s, H, W = f["C"], f["N"], f["s"], f["H"], f["W"]
        hw, ohw = H * W, self.Ho * self.Wo
        kw = dict(ksize=3, stride=s, geom=(b1 - b0, H, W, self.Ho, self.Wo, Cc), scale=self.scale, shift=self.shift, act=L.ACT_SILU)
        if self.res is not None:
            kw.update(R=self.res[b0 * hw:b1 * hw])
        if f.get("post") and post:
            kw.update(post=(self.w2, self.scale2, self.shift2, L.ACT_SILU))
        ops.gemm(self.x[b0 * hw:b1 * hw], self.w, N, 9 * Cc, out=out[b0 * ohw:b1 * ohw], **kw)

    def run(self):
        out = torch.full((self.f["B"] * self.Ho * self.Wo, self.f["N"]), 7.0, device=self.dev, dtype=self.dt)
        self._launch(out, 0, self.f["B"])
        return out

    def reference(self):
        B, N = self.f["B"], self.f["N"]
        out = torch.full((B * self.Ho * self.Wo, N), 7.0, device=self.dev, dtype=self.dt)
        for b in range(B):
            self._launch(out, b, b + 1, post=False)
        if self.f.get("post"):
            two = torch.empty_like(out)
            step = 60000
            for r0 in range(0, out.shape[0], step):
                ops.gemm(out[r0:r0 + step], self.w2, N, N, out=two[r0:r0 + step], scale=self.scale2, shift=self.shift2, act=L.ACT_SILU)
            out = two
        torch.cuda.synchronize()
        return out

    @staticmethod
    def where(got, ref):
        return first_diff(got, ref)

    def base_run(self, tiled=None):
        """The 3x3 ring kernels sum the nine taps in another order than the tiled implicit GEMM (fragment rows are shared between tap
        rows), so they equal it up to the fp32 summation order, not bit for bit: the bit-exact reference of a form is its OWN run on a
        quiet device, accepted only if it sits within two units in the last place of the tiled kernel's result everywhere (a tile read
        before its DMA landed is wrong by whole values, not by ulps)."""
        torch.cuda.synchronize()
        base = self.run()
        torch.cuda.synchronize()
        ref = (tiled if tiled is not None else self.reference()).float()
        rel = 2.0 ** -7 if self.dt == torch.bfloat16 else 2.0 ** -10
        err = (base.float() - ref).abs()
        ok = bool((err <= 2 * rel * ref.abs() + 4 * rel).all())     # (+ an absolute floor: values near zero, and the second product of the POST form)
        return base, ok, float(err.max())


def conv_case(form, dt, dev="cuda"):
    return _ConvCase(form, dt, dev)


GEMM_FORMS = ["value_k128_planes_remap", "value_k256_planes_remap", "value_k256_planes", "n256_k128", "n256_k256_remap", "n256_k384",
              "n512_k512", "n128_k128", "n128_k192", "n128_k256", "seeded_n256_k256", "seeded_n128_k128"]
CONV_FORMS = ["c32", "c32_res", "c64", "c64_res", "c128", "c128_res", "s2_32_64", "s2_64_128", "s2_64_128_post"]


# ------------------------------------------------------------------------------------------------ plan-level checks
def value_planes_vs_tiled(eng, n_frames: int | None = None):
    """The value planes of the first `n_frames` frames as the plan left them, against the SAME value-projection launches restricted
    to those frames (fewer than 65536 rows -> `moy_gemm` takes the tiled kernel): bit for bit.  Call after a pass + synchronize."""
    launches = getattr(eng, "_value_launches", None)
    if not launches or getattr(eng, "value_planes", None) is None or eng.opt.value_planes != 2:
        return dict(equal=True, skipped="no head-plane value launches in this plan")
    arch, B, S = eng.arch, eng.B, getattr(eng, "value_tokens", eng.S)      # (tokens per frame IN THE PLANES: without level 0 when it is sampled raw)
    P, dh = arch.ndl * arch.nh, arch.hd // arch.nh
    rows_max = max(r for _, r in launches)
    n = n_frames or max(1, min(4, B, 60000 // rows_max))
    if n * rows_max >= 65536:
        return dict(equal=True, skipped=f"a level of {rows_max} rows per frame does not fit a tiled launch")
    scr = torch.full((P, n * S, dh), float("nan"), device=eng.dev, dtype=eng.dtype)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    base = eng.value_planes.data_ptr()
    for a_full, rows in launches:
        a = L.GemmArgs()
        C.memmove(C.byref(a), C.byref(a_full), C.sizeof(L.GemmArgs))
        a.M = n * rows
        a.C = scr.data_ptr() + (a_full.C - base)                 # same first row inside a plane (the level's token offset)
        a.plane_stride = n * S * dh
        L.check(eng.lib.moy_gemm(C.byref(a), st), "moy_gemm (value projection, tiled reference)")
    torch.cuda.synchronize()
    got = eng.value_planes.view(P, B, S, dh)[:, :n]
    ref = scr.view(P, n, S, dh)
    eq = bool(torch.equal(got, ref))
    out = dict(equal=eq, frames=n, planes=P, rows_per_frame=S)
    if not eq:
        bad = (got != ref)
        idx = bad.nonzero()[0].tolist()
        out.update(values_differing=int(bad.sum()), first=dict(plane=idx[0], frame=idx[1], token=idx[2], channel=idx[3]),
                   max_abs=float((got.float() - ref.float()).nan_to_num(1e30).abs().max()))
    return out


def _snapshot(eng):
    """Everything a pass leaves behind that another pass must reproduce: outputs, value planes, all token scores, layer views."""
    snap = {k: v.clone() for k, v in eng.outputs().items() if isinstance(v, torch.Tensor)}
    if getattr(eng, "value_planes", None) is not None:
        snap["value_planes"] = eng.value_planes.clone()
    snap["scores_all"] = eng.scores_all.clone()
    for i, v in getattr(eng, "layer_views", {}).items():
        if v is not None and i not in getattr(eng, "virtual_layers", ()):
            snap[f"layer{i}"] = v.tensor().clone()
    return snap


def plan_determinism(eng, passes: int = 3, slot: int = 0, hog: Hog | None = None, eager: bool = False):
    """Run the plan `passes` times on the same input slot and compare every pass with the first, bit for bit.  Returns the list of
    (pass, tensor name, values differing); [] = deterministic.  `eager` replays launch by launch instead of the captured graph."""
    def one():
        if hog is not None:
            hog.kick(8)
        if eager:
            eng.run_steps(slot=slot)
        else:
            eng.forward(slot=slot)
        torch.cuda.synchronize()
        return _snapshot(eng)
    first = one()
    bad = []
    for p in range(1, passes):
        thrash(eng.dev)
        cur = one()
        for k, v in cur.items():
            same = torch.equal(v, first[k]) if not v.is_floating_point() else bool(((v == first[k]) | (v.isnan() & first[k].isnan())).all())
            if not same:
                bad.append(dict(pass_=p, tensor=k, values_differing=int((v != first[k]).sum())))
    return bad, first


def engine_determinism(cfg_name: str, dtype, batch: int, passes: int = 4, device="cuda", hog: bool = True, frames=None):
    """A FRESH engine at `batch` frames: every activation buffer poisoned with NaN, then `passes` passes on the same frames.  Reports
    NaNs in the outputs (an uninitialised read), passes that differ from the first, and the value planes against the tiled kernel."""
    import numpy as np
    from .engine import TrackEngine
    from .fixtures import fixture
    from .synth import SyntheticSequence
    cfg, arch, sd = fixture(cfg_name)
    eng = TrackEngine(arch, sd, cfg["H"], cfg["W"], batch=batch, dtype=dtype, device=device)
    if frames is None:
        seq = SyntheticSequence(0, cfg["H"], cfg["W"], cfg["style"])
        frames = torch.from_numpy(np.concatenate([seq.frames(t0, min(100, batch - t0)) for t0 in range(0, batch, 100)])).to(device)
    eng.inputs[0].copy_(frames)
    n_poisoned = eng.poison_activations()
    res = dict(engine=f"{cfg_name} {str(dtype).replace('torch.', '')} B={batch}", folded_head=bool(getattr(eng, "fold_proj", False)),
               buffers_poisoned=n_poisoned, passes=passes)
    if hog:
        with Hog(device) as h:
            bad, first = plan_determinism(eng, passes, hog=h, eager=True)
    else:
        bad, first = plan_determinism(eng, passes, eager=True)
    res["mismatches"] = bad
    res["nan_outputs"] = [k for k, v in first.items() if v.is_floating_point() and bool(v.isnan().any())]
    res["value_planes_vs_tiled"] = value_planes_vs_tiled(eng)
    del eng
    return res

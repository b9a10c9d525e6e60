"""Race screen of the persistent LDS-DMA ring kernels (round 5).

Round 4 saw the value planes of a fresh engine wrong ONCE (3.26 on a maximum of 12.2).  Cause (DESIGN.md section 4, round 5
item 1): the end-of-tile `s_waitcnt vmcnt(N)` of `gemm_wreg_kernel`, `conv_ws_kernel` and `conv_s2_kernel` used the steady-state
count from the first tile on; at the end of tile 0 there is no store group of a tile -1 in the queue, so the count exceeded what
was outstanding and tile 1 was read without any wait covering its LDS-DMA.  It only went wrong when that DMA took longer than
the whole first tile -- a cold translation, a loaded memory system.  These tests run every ring form of the library again and
again with a bandwidth hog on a second stream and caches thrashed in between, and compare BIT FOR BIT with the tiled
`gemm_kernel` (the same rows as launches below the persistent kernels' thresholds): the forms are bit-identical by
construction, so any difference is a bug, no tolerance.  The engine half: every activation buffer poisoned with NaN before the
first pass (an uninitialised read shows up at once), then repeated passes of the bench-scale plan must reproduce the first one
bit for bit -- value planes, all token scores, every layer view, all outputs.

`tools/stress_rings.py` is the long form of the same screen (thousands of runs, A/B of two library builds via MOYOLO_LIB).
"""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from mo_yolo_amd import _lib as L
from mo_yolo_amd import ops
from mo_yolo_amd.stress import (Hog, conv_case, engine_determinism, gemm_case, thrash)

DEV = "cuda"
REPS = int(os.environ.get("MOY_STRESS_REPS", "12"))


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("form", ["value_k128_planes_remap", "value_k256_planes_remap", "value_k256_planes", "n256_k128", "n256_k256_remap",
                                  "n256_k384", "n512_k512", "n128_k128", "n128_k192", "n128_k256", "seeded_n256_k256", "seeded_n128_k128"])
def test_ring_gemm_forms_bit_identical_to_tiled_under_load(dt, form):
    case = gemm_case(form, dt, DEV)
    ref = case.reference()                       # tiled kernel: launches below 65536 rows
    with Hog(DEV) as hog:
        for i in range(REPS):
            thrash(DEV)
            hog.kick(4)
            got = case.run()
            torch.cuda.synchronize()
            assert torch.equal(got, ref), f"{form} {dt}: run {i} differs from the tiled kernel at {case.where(got, ref)}"


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("form", ["c32", "c32_res", "c64", "c64_res", "c128", "c128_res", "s2_32_64", "s2_64_128", "s2_64_128_post"])
def test_ring_conv_forms_reproduce_their_quiet_run_under_load(dt, form):
    case = conv_case(form, dt, DEV)
    base, close, err = case.base_run()           # quiet run, held to the tiled kernel (per-image launches) within 2 ulp
    assert close, f"{form} {dt}: quiet run is {err} away from the tiled kernel"
    with Hog(DEV) as hog:
        for i in range(REPS):
            thrash(DEV)
            hog.kick(4)
            got = case.run()
            torch.cuda.synchronize()
            assert torch.equal(got, base), f"{form} {dt}: run {i} differs from the quiet run at {case.where(got, base)}"


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_bench_scale_plan_poisoned_buffers_then_bit_identical_passes(dt):
    """B = 104: the smallest batch at which every level takes the folded head and the weight-stationary kernels (B * 19 * 34 >= 65536)."""
    r = engine_determinism("c2", dt, batch=104, passes=6, device=DEV, hog=True)
    assert r["nan_outputs"] == [], r
    assert r["mismatches"] == [], r
    assert r["value_planes_vs_tiled"]["equal"], r["value_planes_vs_tiled"]

"""GPU parity of every C-ABI entry point against the CPU oracle / a plain torch fp32 reference of
the same op, on seeded inputs.  fp32 kernels: tight tolerance; bf16: tolerance stated per test.
Edge cases: ragged M/N (tile tails), K tails, stride 2, image borders, out-of-range sampling
taps, ties and masked tokens in top-k, empty / full ID assignment."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from mo_yolo_amd import _lib as L
from mo_yolo_amd import ops
from oracle import track_oracle as O
from tests._util import golden

DEV = "cuda"
DT = [torch.float32, torch.bfloat16, torch.float16]


def tol(dt, f32=2e-5, bf=3e-2):
    """fp32: tight; bf16 (8 mantissa bits): `bf`; fp16 (11 bits): bf / 4."""
    return f32 if dt == torch.float32 else (bf if dt == torch.bfloat16 else bf / 4)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def q(t, dt):
    """Quantise a CPU fp32 tensor through dtype `dt` (so the reference sees the same inputs)."""
    return t.to(dt).float()


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("M,N,K", [(300, 256, 256), (77, 288, 256), (1000, 64, 96), (129, 1024, 256), (513, 32, 384),
                                   (64, 8, 16)])
def test_gemm_linear_variants(dt, M, N, K):
    x, w = q(rnd(M, K, seed=1), dt), q(rnd(N, K, seed=2, scale=1 / math.sqrt(K)), dt)
    b = rnd(N, seed=3, scale=0.1)
    r = q(rnd(M, N, seed=4), dt)
    xd, wd = x.to(DEV, dt), ops.pad_weight(w.to(DEV), dt)
    # bias + relu
    y = ops.gemm(xd, wd, N, K, shift=b.to(DEV), act=L.ACT_RELU)
    ref = F.relu(x @ w.T + b)
    assert torch.allclose(y.float().cpu(), ref, atol=tol(dt), rtol=tol(dt, 1e-5, 1e-2))
    # scale + shift + silu + residual, fp32 output
    sc = rnd(N, seed=5) * 0.2 + 1.0
    y = ops.gemm(xd, wd, N, K, scale=sc.to(DEV), shift=b.to(DEV), act=L.ACT_SILU, R=r.to(DEV, dt), out_f32=True)
    ref = F.silu((x @ w.T) * sc + b) + r
    assert y.dtype == torch.float32
    assert torch.allclose(y.cpu(), ref, atol=tol(dt, 2e-5, 2e-2), rtol=1e-5)


@pytest.mark.parametrize("M,N,K,scale", [(300, 256, 256, 1.0), (1000, 64, 96, 30.0), (129, 1024, 256, 1e-3), (4097, 32, 384, 1.0), (70000, 256, 128, 1.0),
                                         (66001, 64, 96, 1.0), (65537 + 77, 136, 288, 3.0)])      # (round 6: the 256 x 64 / 256 x 128 tiles, ragged)
def test_gemm_split_f16_products_carry_fp32_accuracy(M, N, K, scale):
    """MOY_F32X3 (round 5): fp32 tensors, every product on the 16-bit matrix cores as hi.hi + (hi.lo + lo.hi) * 2^-11 with
    hi = fp16(x), lo = fp16((x - hi) * 2^11).  Against a float64 product of the same fp32 operands: the error must be of the order of
    the exact-fp32 kernel's own (v_mfma_f32_16x16x4_f32, the parity path) -- a few 2^-22 of the row's magnitude, nowhere near fp16's
    2^-11 --, for operands of very different magnitudes, with bias / SiLU / residual, and through the LayerNorm epilogue."""
    x, w = rnd(M, K, seed=1, scale=scale), rnd(N, K, seed=2, scale=1 / math.sqrt(K))
    b, r = rnd(N, seed=3, scale=0.1 * scale), rnd(M, N, seed=4, scale=scale)
    xd, wd, w3 = x.to(DEV), ops.pad_weight(w.to(DEV), torch.float32), ops.split_weight(w.to(DEV))
    ref = (x.double() @ w.double().T + b.double())
    mag = float((x.abs().double() @ w.abs().double().T).max())            # sum |x||w|: the scale rounding errors are relative to
    y3 = ops.gemm(xd, w3, N, K, shift=b.to(DEV), split_f16=True).double().cpu()
    y1 = ops.gemm(xd, wd, N, K, shift=b.to(DEV)).double().cpu()
    e3, e1 = float((y3 - ref).abs().max()) / mag, float((y1 - ref).abs().max()) / mag
    assert e3 <= 8 * 2.0 ** -22 and e3 <= max(16 * e1, 2.0 ** -21), (e3, e1)
    assert e3 < 2.0 ** -14 / 8                                              # (fp16 products alone would sit at ~2^-11)
    # epilogue forms on top of the split product: scale + shift + SiLU + residual, fp32 in and out
    sc = rnd(N, seed=5) * 0.2 + 1.0
    y = ops.gemm(xd, w3, N, K, scale=sc.to(DEV), shift=b.to(DEV), act=L.ACT_SILU, R=r.to(DEV), split_f16=True)
    want = F.silu((x.double() @ w.double().T) * sc.double() + b.double()) + r.double()
    assert float((y.double().cpu() - want).abs().max()) <= 16 * 2.0 ** -22 * max(mag, scale)
    if N == 256:
        g, be = rnd(N, seed=6) * 0.2 + 1.0, rnd(N, seed=7, scale=0.1)
        y = ops.gemm(xd, w3, N, K, shift=b.to(DEV), R=r.to(DEV), ln=(g.to(DEV), be.to(DEV)), split_f16=True)
        want = F.layer_norm(ref + r.double(), (N,), g.double(), be.double(), 1e-5)
        assert float((y.double().cpu() - want).abs().max()) <= 2e-5
    with pytest.raises(AssertionError):
        ops.gemm(xd.to(torch.bfloat16), ops.pad_weight(w.to(DEV), torch.bfloat16), N, K, split_f16=True)      # a mode of fp32 tensors only


@pytest.mark.parametrize("B,H,W,Cin,Cout,s", [(2, 13, 21, 16, 32, 1), (1, 38, 68, 64, 64, 2), (3, 8, 12, 32, 24, 1),
                                              (2, 190, 190, 64, 64, 1), (1, 261, 259, 16, 128, 1), (2, 366, 370, 32, 64, 2)])   # (the 256-row tiles)
def test_gemm_split_f16_conv3x3(B, H, W, Cin, Cout, s):
    """The implicit-GEMM 3x3 convolution in split precision against float64 (conv.py:36-38: conv + BN + SiLU)."""
    x, w = rnd(B, Cin, H, W, seed=1), rnd(Cout, Cin, 3, 3, seed=2, scale=1 / math.sqrt(9 * Cin))
    sc, sh = rnd(Cout, seed=3) * 0.2 + 1, rnd(Cout, seed=4, scale=0.1)
    ref = F.silu(F.conv2d(x.double(), w.double(), None, s, 1) * sc.double()[None, :, None, None] + sh.double()[None, :, None, None])
    Ho, Wo = ref.shape[2:]
    xin = x.permute(0, 2, 3, 1).reshape(B * H * W, Cin).contiguous().to(DEV)
    wp = ops.split_weight(w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).to(DEV))
    y = ops.gemm(xin, wp, Cout, 9 * Cin, ksize=3, stride=s, geom=(B, H, W, Ho, Wo, Cin), scale=sc.to(DEV), shift=sh.to(DEV), act=L.ACT_SILU,
                 split_f16=True)
    got = y.double().cpu().view(B, Ho, Wo, Cout).permute(0, 3, 1, 2)
    assert float((got - ref).abs().max()) <= 2e-6, float((got - ref).abs().max())


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,act", [(70000, 256, "none"), (65537, 512, "silu"), (66000 + 63, 1536, "none")])
def test_gemm_weight_stationary_kernel_bit_identical_to_tiled(dt, M, N, act):
    """K == 256, M >= 65536, 16-bit: moy_gemm runs the weight-stationary kernel (gemm_wreg.hip).  It must agree with the
    torch fp32 reference and be BIT-identical to the tiled kernel (same MFMA, same k order), which is what runs when the
    same rows are submitted as two launches of fewer than 65536 rows; ragged last tile, partial last row tile, channel-slice
    output (ldc > N) and a padded input pitch (lda > K) included."""
    K = 256
    x, w = q(rnd(M, K, seed=11), dt), q(rnd(N, K, seed=12, scale=1 / math.sqrt(K)), dt)
    b = rnd(N, seed=13, scale=0.1)
    sc = (rnd(N, seed=14) * 0.2 + 1.0) if act == "silu" else None
    xbuf = torch.zeros(M, K + 64, device=DEV, dtype=dt)
    xbuf[:, :K] = x.to(DEV, dt)
    xd = xbuf[:, :K]
    wd = ops.pad_weight(w.to(DEV), dt)
    code = L.ACT_SILU if act == "silu" else L.ACT_NONE
    kw = dict(shift=b.to(DEV), scale=sc.to(DEV) if sc is not None else None, act=code)
    out = torch.full((M + 1, N + 32), 7.0, device=DEV, dtype=dt)       # guard row / guard columns
    ops.gemm(xd, wd, N, K, out=out[:M, :N], **kw)
    h = M // 2
    two = torch.empty(M, N, device=DEV, dtype=dt)
    ops.gemm(xd[:h], wd, N, K, out=two[:h], **kw)
    ops.gemm(xd[h:], wd, N, K, out=two[h:], **kw)
    if N % 512 == 0:   # column planes: every 256-column group as its own contiguous [M, 256] matrix, both kernels
        pl = torch.full((N // 256, M + 1, 256), 7.0, device=DEV, dtype=dt)
        ops.gemm(xd, wd, N, K, out=pl[0, :M], planes=(256, (M + 1) * 256), **kw)
        sm = torch.empty(N // 256, 1000, 256, device=DEV, dtype=dt)
        ops.gemm(xd[:1000], wd, N, K, out=sm[0], planes=(256, 1000 * 256), **kw)
        torch.cuda.synchronize()
        assert torch.equal(pl[:, :M].permute(1, 0, 2).reshape(M, N), out[:M, :N]) and bool((pl[:, M] == 7.0).all())
        assert torch.equal(sm.permute(1, 0, 2).reshape(1000, N), out[:1000, :N])
        hp = torch.full((N // 32, M + 1, 32), 7.0, device=DEV, dtype=dt)      # head planes of 32 columns
        ops.gemm(xd, wd, N, K, out=hp[0, :M], planes=(32, (M + 1) * 32), **kw)
        hs = torch.empty(N // 32, 1000, 32, device=DEV, dtype=dt)
        ops.gemm(xd[:1000], wd, N, K, out=hs[0], planes=(32, 1000 * 32), **kw)
        torch.cuda.synchronize()
        assert torch.equal(hp[:, :M].permute(1, 0, 2).reshape(M, N), out[:M, :N]) and bool((hp[:, M] == 7.0).all())
        assert torch.equal(hs.permute(1, 0, 2).reshape(1000, N), out[:1000, :N])
    torch.cuda.synchronize()
    assert torch.equal(out[:M, :N], two), "weight-stationary kernel differs from the tiled kernel"
    assert bool((out[M] == 7.0).all()) and bool((out[:, N:] == 7.0).all()), "wrote outside its rows / columns"
    ref = x @ w.T
    ref = F.silu(ref * sc + b) if act == "silu" else ref + b
    assert torch.allclose(out[:M, :N].float().cpu(), ref, atol=tol(dt), rtol=tol(dt, 1e-5, 1e-2))


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("K,N,act,rpb", [(128, 256, "silu", 0), (384, 256, "silu", 0), (512, 512, "none", 0), (128, 256, "none", 2584),
                                          (256, 256, "none", 646), (128, 128, "silu", 0), (256, 128, "silu", 0), (192, 128, "silu", 0)])
def test_gemm_weight_stationary_general_k_and_row_remap(dt, K, N, act, rpb):
    """The 32-columns-per-wave forms of the weight-stationary kernel (8 waves: K in {128, 256, 384, 512}, N % 256 == 0; 4 waves:
    N == 128, K in {128, 192, 256}; M >= 65536; 1x1 convs with 256 / 128 outputs and input_proj with its level-major token scatter,
    head.py:1023-1028): bit-identical to the tiled kernel
    (two launches of < 65536 rows), ragged last tile, guard rows / columns untouched."""
    nb = 27 if rpb else 0
    M = nb * rpb if rpb else 66000 + 37
    x, w = q(rnd(M, K, seed=41), dt), q(rnd(N, K, seed=42, scale=1 / math.sqrt(K)), dt)
    b = rnd(N, seed=43, scale=0.1)
    sc = (rnd(N, seed=44) * 0.2 + 1.0) if act == "silu" else None
    xd, wd = x.to(DEV, dt), ops.pad_weight(w.to(DEV), dt)
    kw = dict(shift=b.to(DEV), scale=sc.to(DEV) if sc is not None else None, act=L.ACT_SILU if act == "silu" else L.ACT_NONE)
    if rpb:
        bstride = rpb + 500                                           # rows of other levels in between
        out = torch.full((nb * bstride + 1, N + 32), 7.0, device=DEV, dtype=dt)
        ops.gemm(xd, wd, N, K, out=out[:nb * bstride, :N], c_rpb=rpb, c_bstride=bstride, **kw)
        two = torch.full((nb * bstride, N), 7.0, device=DEV, dtype=dt)
        h = (nb // 2) * rpb
        ops.gemm(xd[:h], wd, N, K, out=two[:(nb // 2) * bstride], c_rpb=rpb, c_bstride=bstride, **kw)
        ops.gemm(xd[h:], wd, N, K, out=two[(nb // 2) * bstride:], c_rpb=rpb, c_bstride=bstride, **kw)
        torch.cuda.synchronize()
        assert torch.equal(out[:nb * bstride, :N], two)
        got = out[:nb * bstride, :N].view(nb, bstride, N)
        assert bool((got[:, rpb:] == 7.0).all()) and bool((out[-1] == 7.0).all()) and bool((out[:, N:] == 7.0).all())
        got = got[:, :rpb].reshape(M, N)
    else:
        out = torch.full((M + 1, N + 32), 7.0, device=DEV, dtype=dt)
        ops.gemm(xd, wd, N, K, out=out[:M, :N], **kw)
        h = M // 2
        two = torch.empty(M, N, device=DEV, dtype=dt)
        ops.gemm(xd[:h], wd, N, K, out=two[:h], **kw)
        ops.gemm(xd[h:], wd, N, K, out=two[h:], **kw)
        torch.cuda.synchronize()
        assert torch.equal(out[:M, :N], two), "weight-stationary kernel differs from the tiled kernel"
        assert bool((out[M] == 7.0).all()) and bool((out[:, N:] == 7.0).all()), "wrote outside its rows / columns"
        got = out[:M, :N]
    ref = x @ w.T
    ref = F.silu(ref * sc + b) if act == "silu" else ref + b
    assert torch.allclose(got.float().cpu(), ref, atol=tol(dt), rtol=tol(dt, 1e-5, 1e-2))


@pytest.mark.parametrize("dt,M,nc", [(torch.bfloat16, 70001, 1), (torch.float16, 66000, 3), (torch.float32, 700, 2),
                                     (torch.bfloat16, 1000, 5)])
def test_gemm_score_only_layernorm_head(dt, M, nc):
    """C == NULL: LayerNorm + narrow head over all rows, rows not stored (enc_score_head over the S tokens, head.py:1036-1042).
    16-bit, M >= 65536, nc <= 4 runs the weight-stationary kernel's one-pass statistics; otherwise the tiled kernel.  Masked
    rows (period mask, wrapping inside tiles) read as zero rows."""
    N = K = 256
    x, w = q(rnd(M, K, seed=21), dt), q(rnd(N, K, seed=22, scale=1 / math.sqrt(K)), dt)
    b, g, be = rnd(N, seed=23, scale=0.1), rnd(N, seed=24) * 0.2 + 1.0, rnd(N, seed=25, scale=0.1)
    dw, db = rnd(nc, N, seed=26, scale=0.1), rnd(nc, seed=27)
    period = 997
    mask = (torch.arange(period) % 5 != 0).to(torch.uint8)
    xd, wd = x.to(DEV, dt), ops.pad_weight(w.to(DEV), dt)
    _, sc = ops.gemm(xd, wd, N, K, shift=b.to(DEV), a_mask=mask.to(DEV), mask_period=period, ln=(g.to(DEV), be.to(DEV)),
                     dot=(dw.to(DEV), db.to(DEV)), store=False)
    mrow = mask[torch.arange(M) % period].float()[:, None]
    ref = F.layer_norm((x * mrow) @ w.T + b, (N,), g, be, 1e-5) @ dw.T + db
    assert sc.shape == (M, nc)
    assert torch.allclose(sc.cpu(), ref, atol=tol(dt, 5e-5, 5e-2))
    # gathered rows carry their token's mask (the recompute of the selected enc_output rows)
    rows = torch.randperm(M, generator=torch.Generator().manual_seed(9))[:300].int()
    y = ops.gemm(xd, wd, N, K, shift=b.to(DEV), a_rows=rows.to(DEV), a_mask=mask.to(DEV), mask_period=period,
                 ln=(g.to(DEV), be.to(DEV)))
    ref = F.layer_norm((x * mrow)[rows.long()] @ w.T + b, (N,), g, be, 1e-5)
    assert torch.allclose(y.float().cpu(), ref, atol=tol(dt, 3e-5, 4e-2))


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("K", [128, 256])
def test_gemm_value_planes_per_level_with_row_remap(dt, K):
    """Round 4: the value projection of ONE pyramid level straight from that level's own [B, h*w, K] tensor (input_proj folded into
    the weights): N = 1536 into head planes of 32 columns WITH the output row remap -- row (b, i) of the level lands at token
    b * S + off + i of every plane (transformer.py:255-257 over head.py:1012-1029).  M >= 65536 takes the weight-stationary value
    form (8 waves x 64 columns; K = 128 is new); it must equal the tiled kernel (the same rows as launches below 65536 rows, whole
    frames each) bit for bit, leave every other row of the planes alone, and agree with torch fp32."""
    B, hw, S, off, N = 70, 1000, 1300, 200, 1536
    M = B * hw
    x, w = q(rnd(M, K, seed=51), dt), q(rnd(N, K, seed=52, scale=1 / math.sqrt(K)), dt)
    b = rnd(N, seed=53, scale=0.1)
    xd, wd = x.to(DEV, dt), ops.pad_weight(w.to(DEV), dt)
    planes = torch.full((N // 32, B * S + 3, 32), 7.0, device=DEV, dtype=dt)
    stride = (B * S + 3) * 32
    ops.gemm(xd, wd, N, K, out=planes[0, off:off + M], shift=b.to(DEV), planes=(32, stride), c_rpb=hw, c_bstride=S)
    two = torch.full((N // 32, B * S + 3, 32), 7.0, device=DEV, dtype=dt)
    per = 60                                                            # 60 000 rows per launch: the tiled kernel
    for b0 in range(0, B, per):
        b1 = min(B, b0 + per)
        ops.gemm(xd[b0 * hw:b1 * hw], wd, N, K, out=two[0, b0 * S + off:b0 * S + off + (b1 - b0) * hw], shift=b.to(DEV),
                 planes=(32, stride), c_rpb=hw, c_bstride=S)
    torch.cuda.synchronize()
    assert torch.equal(planes, two), "value form with row remap differs from the tiled kernel"
    got = planes[:, :B * S].reshape(N // 32, B, S, 32)
    assert bool((got[:, :, :off] == 7.0).all()) and bool((got[:, :, off + hw:] == 7.0).all()) and bool((planes[:, B * S:] == 7.0).all())
    ref = (x @ w.T + b).view(B, hw, N // 32, 32).permute(2, 0, 1, 3)
    assert torch.allclose(got[:, :, off:off + hw].float().cpu(), ref, atol=tol(dt), rtol=tol(dt, 1e-5, 1e-2))


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("K", [128, 256])
def test_gemm_score_pass_per_level_runs_with_own_row_numbering(dt, K):
    """Round 4: the score pass (enc_output + LayerNorm + enc_score_head, head.py:1036-1042, no feature output) of ONE pyramid level
    over the level's valid rectangle, reading that level's own [B, h*w, K] tensor (A row = b * h*w + token - off) while the scores go
    to the level-major [B, S] raster (row b * S + token): `run_a_period` / `run_a_off`.  Against torch fp32; rows outside the run
    keep what the buffer held."""
    B, h, w_, S, off, N, nc = 40, 50, 60, 5000, 700, 256, 1
    hw = h * w_
    y0, y1, x0, x1 = 0, h - 1, 0, 32                                  # the valid rectangle (every row, the first 33 columns)
    x, wt = q(rnd(B * hw, K, seed=61), dt), q(rnd(N, K, seed=62, scale=1 / math.sqrt(K)), dt)
    b, g, be = rnd(N, seed=63, scale=0.1), rnd(N, seed=64) * 0.2 + 1.0, rnd(N, seed=65, scale=0.1)
    dw, db = rnd(nc, N, seed=66, scale=0.1), rnd(nc, seed=67)
    xd, wd = x.to(DEV, dt), ops.pad_weight(wt.to(DEV), dt)
    scores = torch.full((B * S, nc), -5.0, device=DEV)
    runs = dict(period=S, levels=[(off + y0 * w_ + x0, w_, x1 - x0 + 1, y1 - y0 + 1)], a_period=hw, a_off=off)
    ops.gemm(xd, wd, N, K, M=B * S, shift=b.to(DEV), ln=(g.to(DEV), be.to(DEV)), dot=(dw.to(DEV), db.to(DEV)), store=False, runs=runs,
             dot_out=scores)
    torch.cuda.synchronize()
    ref = (F.layer_norm(x @ wt.T + b, (N,), g, be, 1e-5) @ dw.T + db).view(B, h, w_, nc)
    got = scores.cpu().view(B, S, nc)
    lvl = got[:, off:off + hw].view(B, h, w_, nc)
    assert torch.allclose(lvl[:, y0:y1 + 1, x0:x1 + 1], ref[:, y0:y1 + 1, x0:x1 + 1], atol=tol(dt, 5e-5, 5e-2))
    assert bool((lvl[:, :, x1 + 1:] == -5.0).all()) and bool((got[:, :off] == -5.0).all()) and bool((got[:, off + hw:] == -5.0).all())


@pytest.mark.parametrize("dt", DT)
def test_gemm_upsampled_accumulator_seed(dt):
    """Conv1x1(Concat[Upsample2x(u), s]) == GEMM over s seeded with the nearest-2x rows of the half-resolution product W_u.u
    (yolo_track.yaml:28-33 without the upsample / concat copies); ragged tile tails on both GEMMs."""
    B, h, w, cu, cs, N = 2, 7, 9, 64, 32, 96
    H, W = 2 * h, 2 * w
    u, s_ = q(rnd(B, cu, h, w, seed=31), dt), q(rnd(B, cs, H, W, seed=32), dt)
    wt = q(rnd(N, cu + cs, seed=33, scale=1 / math.sqrt(cu + cs)), dt)
    sc, sh = rnd(N, seed=34) * 0.2 + 1, rnd(N, seed=35, scale=0.1)
    cat = torch.cat([F.interpolate(u, scale_factor=2, mode="nearest"), s_], 1)
    ref = F.silu(F.conv2d(cat, wt[:, :, None, None]) * sc[None, :, None, None] + sh[None, :, None, None])
    ud = u.permute(0, 2, 3, 1).reshape(B * h * w, cu).contiguous().to(DEV, dt)
    sd_ = s_.permute(0, 2, 3, 1).reshape(B * H * W, cs).contiguous().to(DEV, dt)
    t = ops.gemm(ud, ops.pad_weight(wt[:, :cu].to(DEV), dt), N, cu, out_f32=True)
    y = ops.gemm(sd_, ops.pad_weight(wt[:, cu:].to(DEV), dt), N, cs, scale=sc.to(DEV), shift=sh.to(DEV), act=L.ACT_SILU,
                 pre=(t, H, W))
    got = y.float().cpu().view(B, H, W, N).permute(0, 3, 1, 2)
    assert torch.allclose(got, ref, atol=tol(dt, 2e-5, 3e-2), rtol=tol(dt, 1e-5, 1e-2))


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("N,K,B,H,W", [(128, 128, 25, 38, 70), (256, 256, 25, 38, 70), (128, 128, 6, 76, 136)])
def test_gemm_seeded_weight_stationary_kernel_bit_identical_to_tiled(dt, N, K, B, H, W):
    """The two seeded 1x1 convs of the neck (yolo_track.yaml:28-33 after the Upsample / Concat fold) at launch sizes that take the
    weight-stationary kernel: the seed rows arrive one tile ahead by hand-counted asynchronous loads.  Bit-identical to the tiled
    kernel (the same rows submitted as launches of whole images below 65536 rows), ragged last tile, output into a channel slice."""
    M, hw = B * H * W, (H // 2) * (W // 2)
    x, w = q(rnd(M, K, seed=41), dt), q(rnd(N, K, seed=42, scale=1 / math.sqrt(K)), dt)
    sc, sh = rnd(N, seed=43) * 0.2 + 1, rnd(N, seed=44, scale=0.1)
    seed = rnd(B * hw, N + 8, seed=45).to(DEV)                      # fp32 half-resolution product, pitch > N
    xd, wd = x.to(DEV, dt), ops.pad_weight(w.to(DEV), dt)
    kw = dict(scale=sc.to(DEV), shift=sh.to(DEV), act=L.ACT_SILU)
    out = torch.full((M + 1, N + 32), 7.0, device=DEV, dtype=dt)
    ops.gemm(xd, wd, N, K, out=out[:M, 16:16 + N], pre=(seed[:, :N], H, W), **kw)
    two = torch.empty(M, N, device=DEV, dtype=dt)
    per = max(1, 65535 // (H * W))
    for b0 in range(0, B, per):
        b1 = min(B, b0 + per)
        ops.gemm(xd[b0 * H * W:b1 * H * W], wd, N, K, out=two[b0 * H * W:b1 * H * W], pre=(seed[b0 * hw:b1 * hw, :N], H, W), **kw)
    torch.cuda.synchronize()
    assert torch.equal(out[:M, 16:16 + N], two), "seeded weight-stationary kernel differs from the tiled kernel"
    assert bool((out[M] == 7.0).all()) and bool((out[:, :16] == 7.0).all()) and bool((out[:, 16 + N:] == 7.0).all())
    up = seed[:, :N].view(B, H // 2, W // 2, N).repeat_interleave(2, 1).repeat_interleave(2, 2).reshape(M, N).cpu()
    ref = F.silu((x @ w.T + up) * sc + sh)
    assert torch.allclose(two.float().cpu(), ref, atol=tol(dt, 2e-5, 3e-2), rtol=tol(dt, 1e-5, 1e-2))


@pytest.mark.parametrize("dt", DT)
def test_gemm_layernorm_residual_prologue_add_gather_mask(dt):
    M, N, K = 333, 256, 256
    x, p = q(rnd(M, K, seed=1), dt), q(rnd(M, K, seed=6), dt)
    w = q(rnd(N, K, seed=2, scale=1 / 16), dt)
    b, g, be = rnd(N, seed=3, scale=0.1), rnd(N, seed=7) * 0.2 + 1, rnd(N, seed=8, scale=0.1)
    r = q(rnd(M, N, seed=4), dt)
    wd = ops.pad_weight(w.to(DEV), dt)
    y = ops.gemm(x.to(DEV, dt), wd, N, K, shift=b.to(DEV), A2=p.to(DEV, dt), R=r.to(DEV, dt), ln=(g.to(DEV), be.to(DEV)))
    xin = q(x + p, dt) if dt == torch.bfloat16 else x + p
    ref = F.layer_norm(xin @ w.T + b + r, (N,), g, be, 1e-5)
    assert torch.allclose(y.float().cpu(), ref, atol=tol(dt, 3e-5, 4e-2))
    # prologue add on the first 512 of 768 columns only (q | k | v of the decoder's self-attention in one launch)
    w3 = q(rnd(768, K, seed=13, scale=1 / math.sqrt(K)), dt)
    y3 = ops.gemm(x.to(DEV, dt), ops.pad_weight(w3.to(DEV), dt), 768, K, A2=p.to(DEV, dt), a2_cols=512)
    ref3 = torch.cat([xin @ w3[:512].T, x @ w3[512:].T], 1)
    assert torch.allclose(y3.float().cpu(), ref3, atol=tol(dt, 3e-5, 4e-2))
    # narrow head fused behind the LayerNorm (enc_score_head on enc_output)
    dw, db = rnd(3, N, seed=11, scale=0.1), rnd(3, seed=12)
    y2, sc = ops.gemm(x.to(DEV, dt), wd, N, K, shift=b.to(DEV), A2=p.to(DEV, dt), R=r.to(DEV, dt), ln=(g.to(DEV), be.to(DEV)),
                      dot=(dw.to(DEV), db.to(DEV)))
    assert torch.equal(y2, y)
    assert torch.allclose(sc.cpu(), ref @ dw.T + db, atol=tol(dt, 5e-5, 5e-2))
    # row gather + row mask (period) + LN
    rows = torch.randperm(M, generator=torch.Generator().manual_seed(9))[:200].int()
    mask = (torch.arange(50) % 3 != 0).to(torch.uint8)
    y = ops.gemm(x.to(DEV, dt), wd, N, K, shift=b.to(DEV), a_rows=rows.to(DEV), ln=(g.to(DEV), be.to(DEV)))
    ref = F.layer_norm(x[rows.long()] @ w.T + b, (N,), g, be, 1e-5)
    assert torch.allclose(y.float().cpu(), ref, atol=tol(dt, 3e-5, 4e-2))
    y = ops.gemm(x.to(DEV, dt), wd, N, K, shift=b.to(DEV), a_mask=mask.to(DEV), mask_period=50)
    mrow = mask[torch.arange(M) % 50].float()[:, None]
    ref = (x * mrow) @ w.T + b
    assert torch.allclose(y.float().cpu(), ref, atol=tol(dt, 2e-5, 2e-2))
    # output row remap (level-major token scatter)
    y = ops.gemm(x[:300].to(DEV, dt), wd, N, K, c_rpb=100, c_bstride=150,
                 out=torch.zeros(450, N, device=DEV, dtype=dt))
    ref = x[:300] @ w.T
    got = y.float().cpu().view(3, 150, N)
    assert torch.allclose(got[:, :100].reshape(300, N), ref, atol=tol(dt, 2e-5, 2e-2))
    assert float(got[:, 100:].abs().max()) == 0.0


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("B,H,W,Cin,Cout,s", [(2, 13, 21, 16, 32, 1), (1, 38, 68, 64, 64, 2), (3, 8, 12, 32, 24, 1),
                                                (1, 19, 34, 128, 128, 1), (2, 16, 16, 8, 16, 2)])
def test_gemm_conv3x3(dt, B, H, W, Cin, Cout, s):
    x = q(rnd(B, Cin, H, W, seed=1), dt)
    w = q(rnd(Cout, Cin, 3, 3, seed=2, scale=1 / math.sqrt(9 * Cin)), dt)
    sc, sh = rnd(Cout, seed=3) * 0.2 + 1, rnd(Cout, seed=4, scale=0.1)
    ref = F.silu(F.conv2d(x, w, None, s, 1) * sc[None, :, None, None] + sh[None, :, None, None])
    Ho, Wo = ref.shape[2:]
    xr = x.permute(0, 2, 3, 1).reshape(B * H * W, Cin).contiguous().to(DEV, dt)
    wp = ops.pad_weight(w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).to(DEV), dt)
    y = ops.gemm(xr, wp, Cout, 9 * Cin, ksize=3, stride=s, geom=(B, H, W, Ho, Wo, Cin), scale=sc.to(DEV), shift=sh.to(DEV),
                 act=L.ACT_SILU)
    got = y.float().cpu().view(B, Ho, Wo, Cout).permute(0, 3, 1, 2)
    assert torch.allclose(got, ref, atol=tol(dt, 2e-5, 3e-2), rtol=1e-5)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("C,B,H,W", [(32, 8, 120, 136), (64, 8, 120, 136), (128, 8, 60, 136)])
@pytest.mark.parametrize("res", [False, True])
def test_conv3x3_weight_stationary_kernel(dt, C, B, H, W, res):
    """Bottleneck convs at launch sizes that take the persistent weight-stationary kernel (csrc/conv_ws.hip: >= 2 tiles per
    CU): image edges that cut tiles in both directions, input / residual / output as channel slices of wider buffers
    (block.py:281-283 `x + cv2(cv1(x))`, conv.py:36-38)."""
    x = q(rnd(B, C, H, W, seed=1), dt)
    w = q(rnd(C, C, 3, 3, seed=2, scale=1 / math.sqrt(9 * C)), dt)
    sc, sh = rnd(C, seed=3) * 0.2 + 1, rnd(C, seed=4, scale=0.1)
    rs = q(rnd(B, C, H, W, seed=5), dt)
    ref = F.silu(F.conv2d(x, w, None, 1, 1) * sc[None, :, None, None] + sh[None, :, None, None])
    if res:
        ref = ref + rs
    M = B * H * W
    buf = torch.zeros(M, 3 * C + 8, device=DEV, dtype=dt)           # [x | residual | out | pad]
    buf[:, :C] = x.permute(0, 2, 3, 1).reshape(M, C).to(DEV, dt)
    buf[:, C:2 * C] = rs.permute(0, 2, 3, 1).reshape(M, C).to(DEV, dt)
    wp = ops.pad_weight(w.permute(0, 2, 3, 1).reshape(C, 9 * C).to(DEV), dt)
    ops.gemm(buf[:, :C], wp, C, 9 * C, ksize=3, stride=1, geom=(B, H, W, H, W, C), scale=sc.to(DEV), shift=sh.to(DEV),
             act=L.ACT_SILU, R=buf[:, C:2 * C] if res else None, out=buf[:, 2 * C:3 * C])
    got = buf[:, 2 * C:3 * C].float().cpu().view(B, H, W, C).permute(0, 3, 1, 2)
    assert torch.allclose(got, ref, atol=tol(dt, 2e-5, 4e-2 if res else 3e-2), rtol=1e-5)
    assert float(buf[:, 3 * C:].abs().max()) == 0                   # nothing written past the output slice
    assert torch.equal(buf[:, :C].float().cpu(), x.permute(0, 2, 3, 1).reshape(M, C))   # input untouched


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("Cin,Cout,B,H,W", [(32, 64, 16, 121, 135), (64, 128, 16, 60, 136), (32, 64, 12, 152, 272)])
def test_conv3x3_stride2_weight_stationary_kernel(dt, Cin, Cout, B, H, W):
    """The down-sampling convs (yolo_track.yaml:18-19,35: 3x3, stride 2, pad 1) at launch sizes that take the persistent
    stride-2 kernel (csrc/conv_ws.hip: parity-de-interleaved patch): odd input sizes, tiles cut by both image edges, input and
    output as channel slices of wider buffers."""
    x = q(rnd(B, Cin, H, W, seed=1), dt)
    w = q(rnd(Cout, Cin, 3, 3, seed=2, scale=1 / math.sqrt(9 * Cin)), dt)
    sc, sh = rnd(Cout, seed=3) * 0.2 + 1, rnd(Cout, seed=4, scale=0.1)
    ref = F.silu(F.conv2d(x, w, None, 2, 1) * sc[None, :, None, None] + sh[None, :, None, None])
    Ho, Wo = ref.shape[2:]
    xin = torch.zeros(B * H * W, Cin + 8, device=DEV, dtype=dt)
    xin[:, :Cin] = x.permute(0, 2, 3, 1).reshape(B * H * W, Cin).to(DEV, dt)
    out = torch.zeros(B * Ho * Wo, Cout + 16, device=DEV, dtype=dt)
    wp = ops.pad_weight(w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).to(DEV), dt)
    ops.gemm(xin[:, :Cin], wp, Cout, 9 * Cin, ksize=3, stride=2, geom=(B, H, W, Ho, Wo, Cin), scale=sc.to(DEV), shift=sh.to(DEV),
             act=L.ACT_SILU, out=out[:, 8:8 + Cout])
    got = out[:, 8:8 + Cout].float().cpu().view(B, Ho, Wo, Cout).permute(0, 3, 1, 2)
    assert torch.allclose(got, ref, atol=tol(dt, 2e-5, 3e-2), rtol=1e-5)
    assert float(out[:, :8].abs().max()) == 0 and float(out[:, 8 + Cout:].abs().max()) == 0


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,H,W", [(16, 60, 136), (3, 151, 271), (40, 152, 272)])
def test_conv3x3_stride2_with_its_1x1_consumer_folded_in(dt, B, H, W):
    """Round 4: Conv(64 -> 128, 3x3, stride 2) whose only consumer is the cv1 of the following C2f (yolo_track.yaml:19-20,
    block.py:225-235): `moy_gemm_args.post_*` applies the 1x1 conv + BN + SiLU to every finished tile on chip, the conv's own
    output never reaches HBM.  BIT-identical to the two launches (the intermediate is rounded where the first launch stored it; the
    1x1 product sums k in the same 32-wide panels as the stand-alone kernels), odd sizes, output as a channel slice of a wider
    buffer (the C2f's concat buffer); and against torch fp32.  A shape without the fused form answers MOY_ENOSYS, it is never
    silently computed without the consumer."""
    Cin, Cout = 64, 128
    x = q(rnd(B, Cin, H, W, seed=1), dt)
    w = q(rnd(Cout, Cin, 3, 3, seed=2, scale=1 / math.sqrt(9 * Cin)), dt)
    w2 = q(rnd(Cout, Cout, seed=5, scale=1 / math.sqrt(Cout)), dt)
    sc, sh = rnd(Cout, seed=3) * 0.2 + 1, rnd(Cout, seed=4, scale=0.1)
    sc2, sh2 = rnd(Cout, seed=6) * 0.2 + 1, rnd(Cout, seed=7, scale=0.1)
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    xin = x.permute(0, 2, 3, 1).reshape(B * H * W, Cin).contiguous().to(DEV, dt)
    wp = ops.pad_weight(w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).to(DEV), dt)
    w2p = ops.pad_weight(w2.to(DEV), dt)
    kw = dict(ksize=3, stride=2, geom=(B, H, W, Ho, Wo, Cin), scale=sc.to(DEV), shift=sh.to(DEV), act=L.ACT_SILU)
    mid = ops.gemm(xin, wp, Cout, 9 * Cin, **kw)
    two = ops.gemm(mid, w2p, Cout, Cout, scale=sc2.to(DEV), shift=sh2.to(DEV), act=L.ACT_SILU)
    cat = torch.full((B * Ho * Wo, 256), 3.0, device=DEV, dtype=dt)
    ops.gemm(xin, wp, Cout, 9 * Cin, out=cat[:, :Cout], post=(w2p, sc2.to(DEV), sh2.to(DEV), L.ACT_SILU), **kw)
    torch.cuda.synchronize()
    assert torch.equal(cat[:, :Cout], two), f"{int((cat[:, :Cout] != two).sum())} of {two.numel()} values differ"
    assert bool((cat[:, Cout:] == 3.0).all())
    m = F.silu(F.conv2d(x, w, None, 2, 1) * sc[None, :, None, None] + sh[None, :, None, None])
    ref = F.silu(F.conv2d(q(m, dt), w2[:, :, None, None]) * sc2[None, :, None, None] + sh2[None, :, None, None])
    got = cat[:, :Cout].float().cpu().view(B, Ho, Wo, Cout).permute(0, 3, 1, 2)
    assert torch.allclose(got, ref, atol=tol(dt, 2e-5, 4e-2), rtol=1e-2)
    with pytest.raises(L.MoyoloError):        # a stride-1 conv has no fused consumer: refused, not ignored
        ops.gemm(mid, ops.pad_weight(q(rnd(Cout, 9 * Cout, seed=8, scale=0.03), dt).to(DEV), dt), Cout, 9 * Cout, ksize=3, stride=1,
                 geom=(B, Ho, Wo, Ho, Wo, Cout), scale=sc.to(DEV), shift=sh.to(DEV), act=L.ACT_SILU,
                 post=(w2p, sc2.to(DEV), sh2.to(DEV), L.ACT_SILU))


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("Cin,Cout,s,B,H,W,res", [(128, 256, 2, 40, 76, 136, False), (256, 256, 2, 156, 38, 68, False),
                                                  (256, 256, 1, 156, 19, 34, True), (192, 256, 1, 10, 77, 135, True),
                                                  (512, 512, 1, 40, 38, 68, False)])
def test_conv3x3_large_tile_dma_kernel_bit_identical_to_tiled(dt, Cin, Cout, s, B, H, W, res):
    """Round 4: the deep 3x3 convolutions (yolo_track.yaml:20-23,35,38; Bottleneck cv2 with its shortcut, block.py:281-283) at launch
    sizes that take the large-tile LDS-DMA kernel (csrc/gemm_dma.hip: K >= 512, N % 256 == 0, >= 384 tiles of 256 x 256).
    BIT-identical to the tiled kernel -- same MFMA, same k order -- which runs when the same images are submitted as launches of
    fewer than 384 tiles; ragged last row tile, tiles that span several images, image borders in both strides, two column tiles,
    a channel count that is not a power of two, input / residual / output as channel slices of wider buffers; a sample of images
    against torch fp32.  (The 512 x 128 and 256 x 128 forms are A/B knobs: test_gemm_dma_other_forms_in_a_child_process.)"""
    x = q(rnd(B, Cin, H, W, seed=1), dt)
    w = q(rnd(Cout, Cin, 3, 3, seed=2, scale=1 / math.sqrt(9 * Cin)), dt)
    sc, sh = rnd(Cout, seed=3) * 0.2 + 1, rnd(Cout, seed=4, scale=0.1)
    Ho, Wo = (H + 2 - 3) // s + 1, (W + 2 - 3) // s + 1
    M = B * Ho * Wo
    bm, bn = (256, 256) if Cout % 256 == 0 else (512, 128)
    assert (M + bm - 1) // bm * (Cout // bn) >= 384
    xin = torch.zeros(B * H * W, Cin + 8, device=DEV, dtype=dt)
    xin[:, :Cin] = x.permute(0, 2, 3, 1).reshape(B * H * W, Cin).to(DEV, dt)
    rs = q(rnd(M, Cout, seed=5), dt).to(DEV, dt) if res else None
    rbuf = None
    if res:
        rbuf = torch.zeros(M, Cout + 4, device=DEV, dtype=dt)
        rbuf[:, 4:] = rs
    out = torch.full((M + 1, Cout + 16), 7.0, device=DEV, dtype=dt)
    wp = ops.pad_weight(w.permute(0, 2, 3, 1).reshape(Cout, 9 * Cin).to(DEV), dt)
    kw = dict(ksize=3, stride=s, scale=sc.to(DEV), shift=sh.to(DEV), act=L.ACT_SILU)
    ops.gemm(xin[:, :Cin], wp, Cout, 9 * Cin, geom=(B, H, W, Ho, Wo, Cin), R=rbuf[:, 4:] if res else None, out=out[:M, 8:8 + Cout], **kw)
    two = torch.empty(M, Cout, device=DEV, dtype=dt)
    per = max(1, 300 * bm // (Ho * Wo) // (Cout // bn))                    # < 384 tiles per launch
    for b0 in range(0, B, per):
        b1 = min(B, b0 + per)
        ops.gemm(xin[b0 * H * W:b1 * H * W, :Cin], wp, Cout, 9 * Cin, geom=(b1 - b0, H, W, Ho, Wo, Cin),
                 R=rbuf[b0 * Ho * Wo:b1 * Ho * Wo, 4:] if res else None, out=two[b0 * Ho * Wo:b1 * Ho * Wo], **kw)
    torch.cuda.synchronize()
    assert torch.equal(out[:M, 8:8 + Cout], two), "large-tile DMA kernel differs from the tiled kernel"
    assert bool((out[M] == 7.0).all()) and bool((out[:, :8] == 7.0).all()) and bool((out[:, 8 + Cout:] == 7.0).all())
    nb = 2
    ref = F.silu(F.conv2d(x[:nb], w, None, s, 1) * sc[None, :, None, None] + sh[None, :, None, None])
    if res:
        ref = ref + rs[:nb * Ho * Wo].float().cpu().view(nb, Ho, Wo, Cout).permute(0, 3, 1, 2)
    got = out[:nb * Ho * Wo, 8:8 + Cout].float().cpu().view(nb, Ho, Wo, Cout).permute(0, 3, 1, 2)
    assert torch.allclose(got, ref, atol=tol(dt, 2e-5, 4e-2 if res else 3e-2), rtol=1e-5)
    last = F.silu(F.conv2d(x[-1:], w, None, s, 1) * sc[None, :, None, None] + sh[None, :, None, None])
    if res:
        last = last + rs[-Ho * Wo:].float().cpu().view(1, Ho, Wo, Cout).permute(0, 3, 1, 2)
    gl = out[M - Ho * Wo:M, 8:8 + Cout].float().cpu().view(1, Ho, Wo, Cout).permute(0, 3, 1, 2)
    assert torch.allclose(gl, last, atol=tol(dt, 2e-5, 4e-2 if res else 3e-2), rtol=1e-5)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K,act,res", [(100001, 512, 1024, "none", False), (100003, 256, 640, "silu", True), (99000, 256, 2048, "relu", False)])
def test_gemm_large_tile_dma_kernel_bit_identical_to_tiled(dt, M, N, K, act, res):
    """The wide 1x1 convolutions of yolo_track.yaml at its own scale (C2f cv2 / SPPF cv2 with K = 640 ... 2048, block.py:129-134,
    178-182) on the 256-row-tile LDS-DMA kernel: bit-identical to the tiled kernel (the same rows as launches below 384 tiles),
    against torch fp32, ragged last tile, padded input pitch, output into a channel slice."""
    x, w = q(rnd(M, K, seed=21), dt), q(rnd(N, K, seed=22, scale=1 / math.sqrt(K)), dt)
    b = rnd(N, seed=23, scale=0.1)
    sc = (rnd(N, seed=24) * 0.2 + 1.0) if act == "silu" else None
    xbuf = torch.zeros(M, K + 64, device=DEV, dtype=dt)
    xbuf[:, :K] = x.to(DEV, dt)
    xd = xbuf[:, :K]
    wd = ops.pad_weight(w.to(DEV), dt)
    r = q(rnd(M, N, seed=25), dt) if res else None
    rd = r.to(DEV, dt) if res else None
    code = {"silu": L.ACT_SILU, "relu": L.ACT_RELU, "none": L.ACT_NONE}[act]
    kw = dict(shift=b.to(DEV), scale=sc.to(DEV) if sc is not None else None, act=code)
    out = torch.full((M + 1, N + 32), 7.0, device=DEV, dtype=dt)
    ops.gemm(xd, wd, N, K, out=out[:M, 8:8 + N], R=rd, **kw)
    two = torch.empty(M, N, device=DEV, dtype=dt)
    step = (256 * 300 // (N // 256)) if N % 256 == 0 else (512 * 300 // (N // 128))
    for m0 in range(0, M, step):
        m1 = min(M, m0 + step)
        ops.gemm(xd[m0:m1], wd, N, K, out=two[m0:m1], R=rd[m0:m1] if res else None, **kw)
    torch.cuda.synchronize()
    assert torch.equal(out[:M, 8:8 + N], two), "large-tile DMA kernel differs from the tiled kernel"
    assert bool((out[M] == 7.0).all()) and bool((out[:, :8] == 7.0).all()) and bool((out[:, 8 + N:] == 7.0).all())
    rows = torch.cat([torch.arange(0, 3000), torch.arange(M - 3000, M)])
    ref = x[rows] @ w.T
    ref = F.silu(ref * sc + b) if act == "silu" else (F.relu(ref + b) if act == "relu" else ref + b)
    if res:
        ref = ref + r[rows]
    assert torch.allclose(out[:M, 8:8 + N][rows.to(DEV)].float().cpu(), ref, atol=tol(dt, 2e-5, 4e-2), rtol=tol(dt, 1e-5, 1e-2))


@pytest.mark.parametrize("form", [1, 2])
def test_gemm_dma_other_forms_in_a_child_process(form):
    """The 512 x 128 (form 1) and 256 x 128 (form 2) tilings of csrc/gemm_dma.hip are measured negatives kept as A/B knobs
    (MOY_GEMM_DMA_FORM, read once per process): one child process per form checks them bit for bit against the tiled kernel
    (tools/probes/gemm_dma_check.py prints the mismatch count per shape)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MOY_GEMM_DMA_FORM=str(form), MOYOLO_LIB=L.lab_library())     # an A/B knob of the LAB library (the product reads no environment)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "probes", "gemm_dma_check.py"), "n128"], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if "mismatches" in ln]
    assert len(lines) >= 3 and all(" mismatches 0 of" in ln for ln in lines), r.stdout[-2000:]


@pytest.mark.parametrize("dtn", ["bf16", "f16"])
def test_conv3x3_grouped_virtual_row_tiling_bit_identical_to_plain_tiling(dtn, tmp_path):
    """Round 4: at C = 128 the weight-stationary 3x3 kernel cuts its 16-column tiles from a VIRTUAL row of G images with one zero
    column between neighbours (68 columns: 5 tiles alone, 13 per three images).  Same arithmetic per pixel, so the outputs must
    equal the plain tiling's (a child process with MOY_CWS_GROUP=0: the switch is read once per process) bit for bit -- batch sizes
    that are not a multiple of the group, odd sizes, the shortcut form, channel-slice operands (the probe checks that nothing
    outside the output slice is written) -- and agree with torch fp32."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools", "probes"))
    import conv_group_check as P
    dt = torch.bfloat16 if dtn == "bf16" else torch.float16
    grouped = P.run(dt)
    out = str(tmp_path / "plain.pt")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "probes", "conv_group_check.py"), out, dtn],
                       env=dict(os.environ, MOY_CWS_GROUP="0", MOYOLO_LIB=L.lab_library()), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    plain = torch.load(out)
    assert len(plain) == len(grouped) == len(P.SHAPES)
    for a, b in zip(grouped, plain):
        assert a["shape"] == b["shape"] and torch.equal(a["x"], b["x"])
        assert torch.equal(a["y"], b["y"]), (a["shape"], int((a["y"] != b["y"]).sum()))
        B, H, W, res = a["shape"]
        x = a["x"].float().view(B, H, W, 128).permute(0, 3, 1, 2)
        w = a["w"].float().view(128, 3, 3, 128).permute(0, 3, 1, 2)
        ref = F.silu(F.conv2d(x, w, None, 1, 1) * a["sc"][None, :, None, None] + a["sh"][None, :, None, None])
        if res:
            ref = ref + a["r"].float().view(B, H, W, 128).permute(0, 3, 1, 2)
        got = a["y"].float().view(B, H, W, 128).permute(0, 3, 1, 2)
        assert torch.allclose(got, ref, atol=tol(dt, 2e-5, 4e-2), rtol=1e-5), float((got - ref).abs().max())


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,H,W", [(2, 64, 96), (3, 100, 132), (1, 608, 1088)])
def test_stem_and_first_downsample_fused(dt, B, H, W):
    """moy_stem_l1_fused = preprocess (BGR->RGB, /255, predictor.py:125-133) + layer 0 + layer 1 (yolo_track.yaml:17-18, Conv + BN +
    SiLU each) against the same chain in torch fp32 with the 16-bit weights; frame sizes that cut tiles at both edges (H/4, W/4
    not multiples of the 8 x 16 tile) and the full C2 frame; output written into a channel slice."""
    g = torch.Generator().manual_seed(0)
    u8 = torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.uint8)
    x = u8.flip(-1).permute(0, 3, 1, 2).float() / 255
    w0 = q(rnd(32, 3, 3, 3, seed=2, scale=0.3), dt)
    w1 = q(rnd(64, 32, 3, 3, seed=5, scale=1 / math.sqrt(288)), dt)
    s0, h0 = rnd(32, seed=3) * 0.2 + 1, rnd(32, seed=4, scale=0.1)
    s1, h1 = rnd(64, seed=6) * 0.2 + 1, rnd(64, seed=7, scale=0.1)
    y0 = F.silu(F.conv2d(x, w0, None, 2, 1) * s0[None, :, None, None] + h0[None, :, None, None])
    ref = F.silu(F.conv2d(q(y0, dt), w1, None, 2, 1) * s1[None, :, None, None] + h1[None, :, None, None])   # layer 0's output is stored in T
    Ho, Wo = H // 4, W // 4
    out = torch.zeros(B * Ho * Wo, 96, device=DEV, dtype=dt)
    ops.stem_l1_fused(u8.to(DEV), ops.stem_weights_fused(w0.to(DEV), dt), s0.to(DEV), h0.to(DEV),
                      ops.pad_weight(w1.permute(0, 2, 3, 1).reshape(64, 288).to(DEV), dt), s1.to(DEV), h1.to(DEV), dt, out=out[:, 16:80])
    got = out[:, 16:80].float().cpu().view(B, Ho, Wo, 64).permute(0, 3, 1, 2)
    assert torch.allclose(got, ref, atol=tol(dt, 1e-5, 4e-2), rtol=1e-5), float((got - ref).abs().max())
    assert float(out[:, :16].abs().max()) == 0 and float(out[:, 80:].abs().max()) == 0


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,H,W", [(2, 19, 45), (1, 152, 272), (3, 8, 30), (2, 9, 61), (1, 272, 480), (40, 40, 272)])
def test_c2f_block_fused(dt, B, H, W):
    """moy_c2f_fused (C2f 64 -> [32 | 32] -> 64, n = 1, shortcut: block.py:219-240, :271-283; conv.py:36-38) against (i) the same
    block as four moy_gemm launches -- equal up to the fp32 summation order inside an MFMA, i.e. an ulp of T on a few values --
    and (ii) torch fp32 with every intermediate rounded to T where the launches store it.  Sizes: tiles cut by both image edges
    (W = 45 -> two tiles of 23; H = 19, 9 -> a last row group of 3 / 1 rows), the C2 and C4 layer-2 geometries (152 x 272 -> 10
    tiles of 28; 272 x 480 -> 16 of 30), exactly one tile; input / output as channel slices of wider buffers; and B x strips
    (40 x 10 = 400) above the grid of one block per CU: blocks then walk WHOLE strips (nfull > 0) and share the
    left-over strips as row ranges (rem > 0) -- the dealing the bench-scale launch (288 frames: nfull 11, rem 64) uses."""
    c = 32
    x = q(rnd(B, 64, H, W, seed=1), dt)
    ws = dict(cv1=q(rnd(64, 64, seed=2, scale=1 / 8), dt), m1=q(rnd(c, c, 3, 3, seed=3, scale=1 / 17), dt),
              m2=q(rnd(c, c, 3, 3, seed=4, scale=1 / 17), dt), cv2=q(rnd(64, 96, seed=5, scale=1 / 10), dt))
    bn = {k: (rnd(n, seed=10 + i) * 0.2 + 1, rnd(n, seed=20 + i, scale=0.1)) for i, (k, n) in enumerate((("cv1", 64), ("m1", c), ("m2", c), ("cv2", 64)))}

    def act(v, k):
        return F.silu(v * bn[k][0][None, :, None, None] + bn[k][1][None, :, None, None])
    y01 = q(act(F.conv2d(x, ws["cv1"][:, :, None, None]), "cv1"), dt)
    y1 = y01[:, c:]
    z = q(act(F.conv2d(y1, ws["m1"], None, 1, 1), "m1"), dt)
    y2 = q(y1 + act(F.conv2d(z, ws["m2"], None, 1, 1), "m2"), dt)
    ref = act(F.conv2d(torch.cat([y01, y2], 1), ws["cv2"][:, :, None, None]), "cv2")

    M = B * H * W
    xin = torch.zeros(M, 80, device=DEV, dtype=dt)
    xin[:, 8:72] = x.permute(0, 2, 3, 1).reshape(M, 64).to(DEV, dt)
    d = lambda t: t.to(DEV)
    wp = dict(cv1=ops.pad_weight(d(ws["cv1"]), dt), m1=ops.pad_weight(d(ws["m1"].permute(0, 2, 3, 1).reshape(c, 9 * c)), dt),
              m2=ops.pad_weight(d(ws["m2"].permute(0, 2, 3, 1).reshape(c, 9 * c)), dt), cv2=ops.pad_weight(d(ws["cv2"]), dt))
    arg = {k: (wp[k], d(bn[k][0]), d(bn[k][1])) for k in wp}
    out = torch.zeros(M, 96, device=DEV, dtype=dt)
    ops.c2f_fused(xin[:, 8:72], B, H, W, arg["cv1"], arg["m1"], arg["m2"], arg["cv2"], out=out[:, 16:80])
    got = out[:, 16:80].float().cpu()
    assert float(out[:, :16].abs().max()) == 0 and float(out[:, 80:].abs().max()) == 0
    assert torch.allclose(got.view(B, H, W, 64).permute(0, 3, 1, 2), ref, atol=tol(dt, 1e-5, 4e-2), rtol=1e-5)

    # the four-launch path on the same buffers
    cat = torch.zeros(M, 96, device=DEV, dtype=dt)
    tmp = torch.zeros(M, c, device=DEV, dtype=dt)
    kw = lambda k: dict(scale=arg[k][1], shift=arg[k][2], act=L.ACT_SILU)
    ops.gemm(xin[:, 8:72], wp["cv1"], 64, 64, out=cat[:, :64], **kw("cv1"))
    ops.gemm(cat[:, c:2 * c], wp["m1"], c, 9 * c, ksize=3, stride=1, geom=(B, H, W, H, W, c), out=tmp, **kw("m1"))
    ops.gemm(tmp, wp["m2"], c, 9 * c, ksize=3, stride=1, geom=(B, H, W, H, W, c), R=cat[:, c:2 * c], out=cat[:, 2 * c:], **kw("m2"))
    four = ops.gemm(cat, wp["cv2"], 64, 96, **kw("cv2")).float().cpu()
    diff = (got - four).abs()
    ulp = 2.0 ** (-8 if dt == torch.bfloat16 else -11)
    assert float((diff / four.abs().clamp_min(0.25)).max()) <= 2 * ulp, float(diff.max())
    assert float((diff > 0).float().mean()) < 0.02                       # all but a few values are bit-identical


@pytest.mark.parametrize("dt", DT)
def test_gemm_channel_slice_views(dt):
    """A / R / C as channel slices of wider concat buffers (C2f / Concat without copies)."""
    M, Cb = 500, 96
    buf = q(rnd(M, Cb, seed=1), dt)
    w = q(rnd(32, 32, seed=2, scale=0.2), dt)
    bd = buf.to(DEV, dt)
    out = torch.zeros(M, 128, device=DEV, dtype=dt)
    ops.gemm(bd[:, 32:64], ops.pad_weight(w.to(DEV), dt), 32, 32, R=bd[:, 64:96], out=out[:, 64:96])
    ref = buf[:, 32:64] @ w.T + buf[:, 64:96]
    assert torch.allclose(out[:, 64:96].float().cpu(), ref, atol=tol(dt, 2e-5, 2e-2))
    assert float(out[:, :64].abs().max()) == 0 and float(out[:, 96:].abs().max()) == 0


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("fmt", ["u8", "f32"])
def test_stem_conv_fused_preprocess(dt, fmt):
    B, H, W, Cout = 2, 32, 48, 16
    g = torch.Generator().manual_seed(0)
    u8 = torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.uint8)
    x = u8.flip(-1).permute(0, 3, 1, 2).float() / 255          # predictor.py:125-133
    w = rnd(Cout, 3, 3, 3, seed=2, scale=0.3)
    sc, sh = rnd(Cout, seed=3) * 0.2 + 1, rnd(Cout, seed=4, scale=0.1)
    ref = F.silu(F.conv2d(x, w, None, 2, 1) * sc[None, :, None, None] + sh[None, :, None, None])
    w27 = w.permute(2, 3, 1, 0).reshape(27, Cout).contiguous().to(DEV)
    src = u8.to(DEV) if fmt == "u8" else x.contiguous().to(DEV)
    y = ops.stem_conv(src, w27, sc.to(DEV), sh.to(DEV), dt)
    got = y.float().cpu().view(B, H // 2, W // 2, Cout).permute(0, 3, 1, 2)
    assert torch.allclose(got, ref, atol=tol(dt, 1e-5, 2e-2))


@pytest.mark.parametrize("B,H,W,Cout", [(2, 32, 48, 16), (1, 608, 1088, 32), (3, 34, 70, 32), (1, 18, 66, 64)])
def test_stem_conv_mfma(B, H, W, Cout):
    """Matrix-core stem (bf16): odd tile tails, image borders, every dword misalignment of the u8 rows."""
    g = torch.Generator().manual_seed(B + W)
    u8 = torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.uint8)
    x = u8.flip(-1).permute(0, 3, 1, 2).float() / 255
    w = q(rnd(Cout, 3, 3, 3, seed=2, scale=0.3), torch.bfloat16)
    sc, sh = rnd(Cout, seed=3) * 0.2 + 1, rnd(Cout, seed=4, scale=0.1)
    ref = F.silu(F.conv2d(q(x, torch.bfloat16), w, None, 2, 1) * sc[None, :, None, None] + sh[None, :, None, None])
    y = ops.stem_conv_mfma(u8.to(DEV), ops.stem_weights_mfma(w.to(DEV)), sc.to(DEV), sh.to(DEV))
    got = y.float().cpu().view(B, H // 2, W // 2, Cout).permute(0, 3, 1, 2)
    assert torch.allclose(got, ref, atol=2e-2), float((got - ref).abs().max())


@pytest.mark.parametrize("B,H,W,Cout", [(2, 32, 48, 16), (1, 608, 1088, 32), (3, 34, 72, 32), (1, 18, 68, 64), (2, 50, 132, 32)])
def test_stem_conv_split_f16(B, H, W, Cout):
    """Round 6: the stem of the f32x3 engine (moy_stem_conv_x3): the bytes as exact fp16 values against the split weights, two products per
    tile -- against float64 (preprocess predictor.py:125-133 + Conv/BN/SiLU conv.py:36-38) at fp32 accuracy, and against the scalar fp32
    stem it replaces in that plan; odd tile tails, image borders, every dword misalignment of the u8 rows."""
    g = torch.Generator().manual_seed(B + W)
    u8 = torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.uint8)
    x = u8.flip(-1).permute(0, 3, 1, 2).double() / 255
    w = rnd(Cout, 3, 3, 3, seed=2, scale=0.3)
    sc, sh = rnd(Cout, seed=3) * 0.2 + 1, rnd(Cout, seed=4, scale=0.1)
    ref = F.silu(F.conv2d(x, w.double(), None, 2, 1) * sc.double()[None, :, None, None] + sh.double()[None, :, None, None])
    y = ops.stem_conv_x3(u8.to(DEV), ops.stem_weights_x3(w.to(DEV)), sc.to(DEV), sh.to(DEV))
    assert y.dtype == torch.float32
    got = y.double().cpu().view(B, H // 2, W // 2, Cout).permute(0, 3, 1, 2)
    e3 = float((got - ref).abs().max())
    y1 = ops.stem_conv(u8.to(DEV), w.permute(2, 3, 1, 0).reshape(27, Cout).contiguous().to(DEV), sc.to(DEV), sh.to(DEV), torch.float32)
    e1 = float((y1.double().cpu().view(B, H // 2, W // 2, Cout).permute(0, 3, 1, 2) - ref).abs().max())
    assert e3 <= 2e-6 and e3 <= max(8 * e1, 1e-6), (e3, e1)
    with pytest.raises(RuntimeError):                # rows of whole dwords only (the engine keeps the fp32 stem otherwise)
        ops.stem_conv_x3(u8[:, :, :W - 2].contiguous().to(DEV), ops.stem_weights_x3(w.to(DEV)), sc.to(DEV), sh.to(DEV))


@pytest.mark.parametrize("Cc", [16, 64, 128])
@pytest.mark.parametrize("dt", DT)
def test_sppf_pool_and_upsample(dt, Cc):
    """(Cc >= 32 at 16 bits takes the four-chunks-per-block form of sppf_pool_kernel)"""
    B, H, W = 2, 19, 34
    x = q(rnd(B, Cc, H, W, seed=1), dt)
    xr = x.permute(0, 2, 3, 1).reshape(-1, Cc).contiguous().to(DEV, dt)
    y1 = F.max_pool2d(x, 5, 1, 2); y2 = F.max_pool2d(y1, 5, 1, 2); y3 = F.max_pool2d(y2, 5, 1, 2)
    for got, ref in zip(ops.sppf_pool(xr, B, H, W), (y1, y2, y3)):
        assert torch.equal(got.float().cpu().view(B, H, W, Cc).permute(0, 3, 1, 2), ref)
    up = ops.upsample2x(xr, B, H, W).float().cpu().view(B, 2 * H, 2 * W, Cc).permute(0, 3, 1, 2)
    assert torch.equal(up, F.interpolate(x, scale_factor=2.0, mode="nearest"))


@pytest.mark.parametrize("dt", DT)
def test_rowdot_modes(dt):
    M, K = 257, 256
    x = q(rnd(M, K, seed=1), dt)
    w, b = rnd(4, K, seed=2, scale=0.1), rnd(4, seed=3)
    xd = x.to(DEV, dt)
    y = ops.rowdot(xd, w.to(DEV), b.to(DEV))
    assert torch.allclose(y.cpu(), x @ w.T + b, atol=2e-5)
    w1 = rnd(1, K, seed=5, scale=0.1)
    assert torch.allclose(ops.rowdot(xd, w1.to(DEV), b[:1].to(DEV)).cpu(), x @ w1.T + b[:1], atol=2e-5)
    ref_box = torch.rand(M, 4, generator=torch.Generator().manual_seed(4))
    ref_box[0] = torch.tensor([0.0, 1.0, 1e-7, 1 - 1e-7])      # inverse_sigmoid clamps (eps 1e-5)
    y = ops.rowdot(xd, w.to(DEV), b.to(DEV), mode=1, aux=ref_box.to(DEV))
    assert torch.allclose(y.cpu(), torch.sigmoid(x @ w.T + b + O.inverse_sigmoid(ref_box)), atol=2e-6)
    anchors = rnd(100, 4, seed=6)
    anchors[7] = float("inf")
    ar = (torch.arange(M) * 7 % 100).int()
    y = ops.rowdot(xd, w.to(DEV), b.to(DEV), mode=2, aux=anchors.to(DEV), aux_rows=ar.to(DEV))
    ref = x @ w.T + b + anchors[ar.long()]
    fin = torch.isfinite(ref)
    assert torch.equal(torch.isfinite(y.cpu()), fin) and torch.allclose(y.cpu()[fin], ref[fin], atol=2e-5)
    rows = torch.randperm(M, generator=torch.Generator().manual_seed(9))[:50].int()
    y = ops.rowdot(xd, w.to(DEV), b.to(DEV), x_rows=rows.to(DEV))
    assert torch.allclose(y.cpu(), x[rows.long()] @ w.T + b, atol=2e-5)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,mode", [(300, 1), (1000 + 7, 2), (33, 0)])
def test_mlp_head_one_launch_equals_three(dt, M, mode):
    """moy_mlp_head (box head in one launch) vs moy_gemm x 2 + moy_rowdot: identical hidden activations (same rounding points,
    same k order), the 4 outputs equal up to the fp32 summation order; and vs the torch fp32 MLP."""
    x = q(rnd(M + 50, 256, seed=1), dt)
    W0, W1 = q(rnd(256, 256, seed=2, scale=1 / 16), dt), q(rnd(256, 256, seed=3, scale=1 / 16), dt)
    b0, b1, w2, b2 = rnd(256, seed=4, scale=0.1), rnd(256, seed=5, scale=0.1), rnd(4, 256, seed=6, scale=0.1), rnd(4, seed=7)
    rows = torch.randperm(M + 50, generator=torch.Generator().manual_seed(8))[:M].int() if mode == 2 else None
    aux = torch.rand(M if mode != 2 else 77, 4, generator=torch.Generator().manual_seed(9))
    aux_rows = (torch.arange(M) % 77).int() if mode == 2 else None
    xd, W0d, W1d = x.to(DEV, dt), ops.pad_weight(W0.to(DEV), dt), ops.pad_weight(W1.to(DEV), dt)
    kw = dict(mode=mode, aux=aux.to(DEV) if mode else None, aux_rows=aux_rows.to(DEV) if aux_rows is not None else None)
    xin = xd if rows is not None else xd[:M]
    y = ops.mlp_head(xin, W0d, b0.to(DEV), W1d, b1.to(DEV), w2.to(DEV), b2.to(DEV), x_rows=rows.to(DEV) if rows is not None else None, **kw)
    t1 = ops.gemm(xin, W0d, 256, 256, shift=b0.to(DEV), act=L.ACT_RELU, a_rows=rows.to(DEV) if rows is not None else None)
    t2 = ops.gemm(t1, W1d, 256, 256, shift=b1.to(DEV), act=L.ACT_RELU)
    y3 = ops.rowdot(t2, w2.to(DEV), b2.to(DEV), **kw)
    assert torch.allclose(y, y3, atol=2e-5, rtol=1e-5)
    xs = x[rows.long()] if rows is not None else x[:M]
    ref = F.relu(q(F.relu(xs @ W0.T + b0), dt) @ W1.T + b1)
    ref = q(ref, dt) @ w2.T + b2
    if mode == 1:
        a = aux.clamp(0, 1)
        ref = torch.sigmoid(ref + torch.log(a.clamp(min=1e-5) / (1 - a).clamp(min=1e-5)))
    elif mode == 2:
        ref = ref + aux[aux_rows.long()]
    assert torch.allclose(y.cpu(), ref, atol=tol(dt, 1e-5, 3e-2))


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,dffn", [(300, 1024), (1000 + 7, 512)])
def test_decoder_tail_one_launch_equals_separate(dt, M, dffn):
    """moy_decoder_tail (output_proj + norm2, FFN + norm3, box refinement in one launch) vs the separate launches
    (moy_gemm x 3 + moy_mlp_head) and vs the torch fp32 chain with the same rounding points (transformer.py:642-652, 705-709)."""
    g = lambda *s_, seed, scale=1.0: rnd(*s_, seed=seed, scale=scale)
    samp, e1 = q(g(M, 256, seed=1), dt), q(g(M, 256, seed=2), dt)
    Wp, W1, W2 = q(g(256, 256, seed=3, scale=1 / 16), dt), q(g(dffn, 256, seed=4, scale=1 / 16), dt), q(g(256, dffn, seed=5, scale=1 / 32), dt)
    B0, B1 = q(g(256, 256, seed=6, scale=1 / 16), dt), q(g(256, 256, seed=7, scale=1 / 16), dt)
    bp, b1, b2, c0, c1 = (g(256, seed=8, scale=0.1), g(dffn, seed=9, scale=0.1), g(256, seed=10, scale=0.1), g(256, seed=11, scale=0.1),
                          g(256, seed=12, scale=0.1))
    g2, be2, g3, be3 = g(256, seed=13) * 0.2 + 1, g(256, seed=14, scale=0.1), g(256, seed=15) * 0.2 + 1, g(256, seed=16, scale=0.1)
    w2, c2 = g(4, 256, seed=17, scale=0.1), g(4, seed=18)
    ref_in = torch.rand(M, 4, generator=torch.Generator().manual_seed(19))
    d = lambda t: t.to(DEV)
    pw = lambda w: ops.pad_weight(w.to(DEV), dt)
    sd_, e1d = samp.to(DEV, dt), e1.to(DEV, dt)
    out, ref_out = ops.decoder_tail(sd_, e1d, pw(Wp), d(bp), (d(g2), d(be2)), pw(W1), d(b1), pw(W2), d(b2), (d(g3), d(be3)),
                                    pw(B0), d(c0), pw(B1), d(c1), d(w2), d(c2), d(ref_in))
    # the same call with the five matrices in MFMA-fragment order (include/moyolo.h; what the engine passes): the same bits
    pk = lambda w: ops.pack_mfma_a(pw(w))
    out_p, ref_p = ops.decoder_tail(sd_, e1d, pk(Wp), d(bp), (d(g2), d(be2)), pk(W1), d(b1), pk(W2), d(b2), (d(g3), d(be3)),
                                    pk(B0), d(c0), pk(B1), d(c1), d(w2), d(c2), d(ref_in), packed=True)
    assert torch.equal(out_p, out) and torch.equal(ref_p, ref_out)
    # ... and with the NEXT layer's q | k | v projection on the rows while they are on chip (round 5): the bits of moy_gemm over
    # out / out + query_pos (transformer.py:637-640)
    Wqkv, bqkv, qpos = g(768, 256, seed=21, scale=1 / 16), g(768, seed=22, scale=0.2), g(M, 256, seed=23).to(DEV, dt)
    for pack in (False, True):
        wq = ops.pack_mfma_a(pw(Wqkv)) if pack else pw(Wqkv)
        ws = [pk(w) if pack else pw(w) for w in (Wp, W1, W2, B0, B1)]
        o3, r3_, qkv = ops.decoder_tail(sd_, e1d, ws[0], d(bp), (d(g2), d(be2)), ws[1], d(b1), ws[2], d(b2), (d(g3), d(be3)),
                                        ws[3], d(c0), ws[4], d(c1), d(w2), d(c2), d(ref_in), packed=pack, next_qkv=(wq, d(bqkv), qpos))
        assert torch.equal(o3, out) and torch.equal(r3_, ref_out)
        qk_ref = ops.gemm(out, pw(Wqkv[:512]), 512, 256, shift=d(bqkv[:512]), A2=qpos)
        v_ref = ops.gemm(out, pw(Wqkv[512:]), 256, 256, shift=d(bqkv[512:]))
        assert torch.equal(qkv[:, :512], qk_ref) and torch.equal(qkv[:, 512:], v_ref)
    # separate launches
    e2 = ops.gemm(sd_, pw(Wp), 256, 256, shift=d(bp), R=e1d, ln=(d(g2), d(be2)))
    h = ops.gemm(e2, pw(W1), dffn, 256, shift=d(b1), act=L.ACT_RELU)
    e3 = ops.gemm(h, pw(W2), 256, dffn, shift=d(b2), R=e2, ln=(d(g3), d(be3)))
    r3 = ops.mlp_head(e3, pw(B0), d(c0), pw(B1), d(c1), d(w2), d(c2), mode=1, aux=d(ref_in))
    ulp = 2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10
    # a handful of elements may land on the other side of a rounding boundary (one-pass vs two-pass LayerNorm statistics)
    diff = (out.float() - e3.float()).abs()
    assert float(diff.max()) <= 4 * ulp * float(e3.float().abs().max()) and float((diff > 0).float().mean()) < 0.02
    assert torch.allclose(ref_out, r3, atol=3e-3 if dt == torch.bfloat16 else 5e-4)
    # torch fp32 chain with the storage-type rounding points
    r_e2 = q(F.layer_norm(samp @ Wp.T + bp + e1, (256,), g2, be2, 1e-5), dt)
    r_h = q(F.relu(r_e2 @ W1.T + b1), dt)
    r_e3 = q(F.layer_norm(r_h @ W2.T + b2 + r_e2, (256,), g3, be3, 1e-5), dt)
    assert torch.allclose(out.float().cpu(), r_e3, atol=tol(dt, 1e-5, 4e-2))
    t2 = q(F.relu(q(F.relu(r_e3 @ B0.T + c0), dt) @ B1.T + c1), dt)
    a = ref_in.clamp(0, 1)
    r_box = torch.sigmoid(t2 @ w2.T + c2 + torch.log(a.clamp(min=1e-5) / (1 - a).clamp(min=1e-5)))
    assert torch.allclose(ref_out.cpu(), r_box, atol=tol(dt, 1e-5, 2e-2))


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
def test_decoder_tail_block_heights_are_bit_identical(dt):
    """Round 6: at small M (the small-batch leg) moy_decoder_tail runs 32- or 64-row blocks instead of 128-row ones (a block's chain of
    eleven dependent products is the launch's duration there).  Every row's arithmetic is independent of the block height: the first rows
    of a 24 653-row launch (128-row blocks) == the same rows launched alone as 12 300 rows (64-row blocks) and as 1 000 rows (32-row
    blocks), bit for bit -- outputs, refined boxes and the next layer's q | k | v."""
    M, dffn = 128 * 192 + 77, 1024
    g = lambda *s_, seed, scale=1.0: rnd(*s_, seed=seed, scale=scale)
    samp, e1 = g(M, 256, seed=1).to(DEV, dt), g(M, 256, seed=2).to(DEV, dt)
    pk = lambda w: ops.pack_mfma_a(ops.pad_weight(w.to(DEV), dt))
    Wp, W1, W2 = pk(g(256, 256, seed=3, scale=1 / 16)), pk(g(dffn, 256, seed=4, scale=1 / 16)), pk(g(256, dffn, seed=5, scale=1 / 32))
    B0, B1 = pk(g(256, 256, seed=6, scale=1 / 16)), pk(g(256, 256, seed=7, scale=1 / 16))
    d = lambda t: t.to(DEV)
    vec = [d(g(256, seed=8, scale=0.1)), d(g(dffn, seed=9, scale=0.1)), d(g(256, seed=10, scale=0.1)), d(g(256, seed=11, scale=0.1)), d(g(256, seed=12, scale=0.1))]
    ln2, ln3 = (d(g(256, seed=13) * 0.2 + 1), d(g(256, seed=14, scale=0.1))), (d(g(256, seed=15) * 0.2 + 1), d(g(256, seed=16, scale=0.1)))
    w2, c2 = d(g(4, 256, seed=17, scale=0.1)), d(g(4, seed=18))
    ref_in = torch.rand(M, 4, generator=torch.Generator().manual_seed(19)).to(DEV)
    Wqkv, bqkv, qpos = pk(g(768, 256, seed=21, scale=1 / 16)), d(g(768, seed=22, scale=0.2)), g(M, 256, seed=23).to(DEV, dt)

    def run(n):
        return ops.decoder_tail(samp[:n].contiguous(), e1[:n].contiguous(), Wp, vec[0], ln2, W1, vec[1], W2, vec[2], ln3, B0, vec[3], B1, vec[4], w2, c2,
                                ref_in[:n].contiguous(), packed=True, next_qkv=(Wqkv, bqkv, qpos[:n].contiguous()))
    full = run(M)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(full[0].float()).all())
    for n in (12300, 1000):
        part = run(n)
        torch.cuda.synchronize()
        for a, b in zip(part, full):
            assert torch.equal(a, b[:n]), n


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,n_oa", [(300 * 3 + 7, 288), (128, 288), (1000, 384), (513, 192), (200, 96)])
def test_decoder_mid_one_launch_equals_separate(dt, M, n_oa):
    """moy_decoder_mid (out_proj + norm1, then sampling_offsets | attention_weights of e1 + query_pos, in one launch) vs the two
    moy_gemm launches it replaces and vs the torch fp32 chain with the same rounding points (transformer.py:640-646, :262-266);
    n_oa = 8 heads x levels x 4 points x 3 for 3 / 4 / 2 / 1 levels (the remainder columns past 256 take the row-split product)."""
    g = lambda *s_, seed, scale=1.0: rnd(*s_, seed=seed, scale=scale)
    attn, x, qpos = q(g(M, 256, seed=1), dt), q(g(M, 256, seed=2), dt), q(g(M, 256, seed=3), dt)
    Wo, Woa = q(g(256, 256, seed=4, scale=1 / 16), dt), q(g(n_oa, 256, seed=5, scale=1 / 16), dt)
    bo, boa = g(256, seed=6, scale=0.1), g(n_oa, seed=7, scale=0.5)
    g1, be1 = g(256, seed=8) * 0.2 + 1, g(256, seed=9, scale=0.1)
    d = lambda t: t.to(DEV)
    pw = lambda w: ops.pad_weight(w.to(DEV), dt)
    ad, xd, qd = attn.to(DEV, dt), x.to(DEV, dt), qpos.to(DEV, dt)
    Wpad = torch.zeros(max(256, n_oa), 256)
    Wpad[:n_oa] = Woa
    e1, offaw = ops.decoder_mid(ad, xd, qd, pw(Wo), d(bo), (d(g1), d(be1)), pw(Wpad), d(boa), n_oa)
    e1p, offawp = ops.decoder_mid(ad, xd, qd, ops.pack_mfma_a(pw(Wo)), d(bo), (d(g1), d(be1)), ops.pack_mfma_a(pw(Wpad)), d(boa), n_oa, packed=True)
    assert torch.equal(e1p, e1) and torch.equal(offawp, offaw)          # weights in MFMA-fragment order: the same bits
    # separate launches (the fp32 engines' path)
    e1s = ops.gemm(ad, pw(Wo), 256, 256, shift=d(bo), R=xd, ln=(d(g1), d(be1)))
    oas = ops.gemm(e1s, pw(Woa), n_oa, 256, shift=d(boa), A2=qd, out_f32=True)
    ulp = 2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10
    diff = (e1.float() - e1s.float()).abs()
    # a handful of elements may land on the other side of a rounding boundary (one-pass vs two-pass LayerNorm statistics)
    assert float(diff.max()) <= 4 * ulp * float(e1s.float().abs().max()) and float((diff > 0).float().mean()) < 0.02
    assert torch.allclose(offaw, oas, atol=tol(dt, 1e-5, 3e-2 if dt == torch.bfloat16 else 4e-3))
    # the offsets / weights of the rows whose e1 agrees bit for bit agree to fp32 summation order
    same = (diff.max(1).values == 0)
    assert float(same.float().mean()) > 0.2
    assert torch.allclose(offaw[same], oas[same], atol=2e-5, rtol=1e-5)
    # torch fp32 chain with the storage-type rounding points
    r_e1 = q(F.layer_norm(attn @ Wo.T + bo + x, (256,), g1, be1, 1e-5), dt)
    assert torch.allclose(e1.float().cpu(), r_e1, atol=tol(dt, 1e-5, 4e-2))
    r_oa = q(r_e1 + qpos, dt) @ Woa.T + boa
    assert torch.allclose(offaw.cpu(), r_oa, atol=tol(dt, 1e-5, 6e-2))


@pytest.mark.parametrize("B,S,nc,nq", [(1, 13566, 1, 300), (3, 315, 1, 50), (2, 126, 3, 20), (1, 42840, 1, 500), (2, 1000, 2, 1000)])
def test_topk_matches_torch_and_flags_masked(B, S, nc, nq):
    sc = rnd(B, S, nc, seed=B + S)
    il, ig, nm = ops.topk(sc.to(DEV), nq)
    ref = torch.topk(sc.max(-1).values, nq, dim=1).indices
    assert torch.equal(il.cpu().long(), ref)
    assert torch.equal(ig.cpu().long(), ref + torch.arange(B)[:, None] * S)
    # ties (constant masked-token score): lowest index first, masked selections counted
    sc2 = sc.clone()
    valid = torch.ones(S, dtype=torch.uint8)
    valid[S // 2:] = 0
    sc2[:, S // 2:] = 10.0                                   # masked tokens outrank everything, all tied
    il, _, nm = ops.topk(sc2.to(DEV), nq, valid.to(DEV))
    k = min(nq, S - S // 2)
    assert torch.equal(il.cpu()[:, :k].long(), (S // 2 + torch.arange(k)).expand(B, k))
    assert nm.cpu().tolist() == [k] * B


@pytest.mark.parametrize("dt", DT)
def test_pos2posemb(dt):
    pos = rnd(123, 4, seed=1, scale=6.0)
    y = ops.pos2posemb(pos.to(DEV), dt)
    assert torch.allclose(y.float().cpu(), O.pos2posemb(pos), atol=tol(dt, 3e-5, 1e-2))


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("B,Lq", [(1, 300), (2, 50), (1, 500), (3, 7)])
def test_mha_core(dt, B, Lq):
    E, nh = 256, 8
    qkv = q(rnd(B * Lq, 3 * E, seed=Lq), dt)
    y = ops.mha_core(qkv.to(DEV, dt), B, Lq, nh)
    t = qkv.view(B, Lq, 3, nh, 32)
    qq, kk, vv = (t[:, :, i].transpose(1, 2) for i in range(3))
    a = torch.softmax((qq / math.sqrt(32)) @ kk.transpose(-1, -2), -1)
    ref = (a @ vv).transpose(1, 2).reshape(B * Lq, E)
    assert torch.allclose(y.float().cpu(), ref, atol=tol(dt, 1e-5, 1e-2))


@pytest.mark.parametrize("dt", DT)
def test_msda_fused_vs_oracle(dt):
    B, Lq, shapes = 2, 37, [(12, 20), (6, 10), (3, 5)]
    S = sum(h * w for h, w in shapes)
    value = q(rnd(B, S, 256, seed=1), dt)
    offaw = torch.cat([rnd(B * Lq, 192, seed=2, scale=6.0), rnd(B * Lq, 96, seed=3, scale=2.0)], 1)
    ref_box = torch.rand(B * Lq, 4, generator=torch.Generator().manual_seed(4))
    ref_box[:, 2:] = ref_box[:, 2:] * 0.5 + 0.05
    ref_box[0, :2] = torch.tensor([0.01, 0.99])               # taps fall outside: zero padding
    y = ops.msda_fused(value.view(B * S, 256).to(DEV, dt), B, S, shapes, offaw.to(DEV), ref_box.to(DEV), Lq)
    off = offaw[:, :192].view(B, Lq, 8, 3, 4, 2)
    aw = torch.softmax(offaw[:, 192:].view(B, Lq, 8, 12), -1).view(B, Lq, 8, 3, 4)
    rb = ref_box.view(B, Lq, 1, 1, 1, 4)
    loc = rb[..., :2] + off / 4 * rb[..., 2:] * 0.5
    ref = O.msda_core(value.view(B, S, 8, 32), shapes, loc, aw).view(B * Lq, 256)
    assert torch.allclose(y.float().cpu(), ref, atol=tol(dt, 2e-5, 1e-2))
    # head planes [8][B*S][32] (the layout the value projection writes for the gather).  16-bit: the paired-corner kernel
    # (msda_planes_kernel) sums the two x-corners in separate lanes -- same terms, another fp32 order: equal up to an ulp of T;
    # fp32 keeps the one-corner kernel: bit for bit
    vp = value.view(B * S, 8, 32).permute(1, 0, 2).contiguous().to(DEV, dt)
    y2 = ops.msda_fused(vp, B, S, shapes, offaw.to(DEV), ref_box.to(DEV), Lq, head_planes=True)
    if dt == torch.float32:
        assert torch.equal(y2, y)
    else:
        assert torch.allclose(y2.float().cpu(), ref, atol=tol(dt, 2e-5, 1e-2))
        ulp = 2.0 ** (-8 if dt == torch.bfloat16 else -11)
        assert float(((y2.float() - y.float()).abs() / y.float().abs().clamp_min(0.05)).max()) <= 2 * ulp


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,Lq,shapes,ld0", [(2, 37, [(12, 20), (6, 10), (3, 5)], 128), (3, 300, [(76, 136), (38, 68), (19, 34)], 128),
                                            (1, 21, [(9, 7)], 192), (2, 50, [(2, 2), (4, 4)], 136), (2, 45, [(5, 2), (2, 3), (2, 2)], 128),
                                            (1, 33, [(7, 9), (1, 5), (3, 3)], 128)])
def test_msda_raw_level0_gather_then_project_vs_oracle(dt, B, Lq, shapes, ld0):
    """Round 5: level 0 of the deformable attention gathered RAW and projected after the bilinear sum (csrc/msda_raw.hip):
    W_h . (sum_p a_p bilinear(x)(loc_p)) + c_h . sum_p a_p (in-range corner weights) == sum_p a_p bilinear(W x + c)(loc_p)
    (transformer.py:255-287 over nn/modules/utils.py:41-78, zero padding of the PROJECTED map).  Against the oracle's sampling core on
    the explicitly projected level-0 map (fp32 product of the same 16-bit operands) + the given planes of the other levels; offsets large
    enough that samples leave the level on every side, a box in a corner, one level only, a level of 2 x 2, level 0 as a channel slice.
    Three levels of at least 2 x 2 run the form with the tap sums on the matrix cores (weights enter it as T(w) + T(w - T(w))), every other
    shape (one / two levels, a level one pixel high) the vector-ALU form."""
    H0, W0 = shapes[0]
    S1 = sum(h * w for h, w in shapes[1:])
    nl = len(shapes)
    x = q(rnd(B * H0 * W0, 128, seed=1), dt)
    wc, bc = q(rnd(256, 128, seed=2, scale=1 / math.sqrt(128)), dt), rnd(256, seed=3, scale=0.5)
    v1 = q(rnd(B, max(S1, 1), 256, seed=4), dt)
    offaw = torch.cat([rnd(B * Lq, 8 * nl * 8, seed=5, scale=6.0), rnd(B * Lq, 8 * nl * 4, seed=6, scale=2.0)], 1)
    ref_box = torch.rand(B * Lq, 4, generator=torch.Generator().manual_seed(7))
    ref_box[:, 2:] = ref_box[:, 2:] * 0.5 + 0.05
    ref_box[0, :2] = torch.tensor([0.01, 0.99])               # taps fall outside: zero padding (bias included)
    ref_box[1, :2] = torch.tensor([0.995, 0.002])
    xbuf = torch.zeros(B * H0 * W0, ld0, device=DEV, dtype=dt)
    xbuf[:, :128] = x.to(DEV, dt)
    planes = v1.view(B * S1, 8, 32).permute(1, 0, 2).contiguous().to(DEV, dt) if S1 else None
    y = ops.msda_raw0(xbuf[:, :128], wc.to(DEV, dt).contiguous(), bc.to(DEV), planes, B, shapes, offaw.to(DEV), ref_box.to(DEV), Lq)
    yp = ops.msda_raw0(xbuf[:, :128], ops.pack_mfma_a(wc.to(DEV, dt)), bc.to(DEV), planes, B, shapes, offaw.to(DEV), ref_box.to(DEV), Lq, packed=True)
    torch.cuda.synchronize()
    assert torch.equal(yp, y)                                            # composed weights in MFMA-fragment order: the same bits
    v0 = (x @ wc.T + bc).view(B, H0 * W0, 256)
    value = torch.cat([v0, v1[:, :S1]], 1) if S1 else v0
    off = offaw[:, :8 * nl * 8].view(B, Lq, 8, nl, 4, 2)
    aw = torch.softmax(offaw[:, 8 * nl * 8:].view(B, Lq, 8, nl * 4), -1).view(B, Lq, 8, nl, 4)
    rb = ref_box.view(B, Lq, 1, 1, 1, 4)
    loc = rb[..., :2] + off / 4 * rb[..., 2:] * 0.5
    want = O.msda_core(value.view(B, -1, 8, 32), shapes, loc, aw).view(B * Lq, 256)
    # one rounding of the gathered vector to T before its product (|g| <= 1, 128 terms of |w| ~ 0.09) + the output rounding
    assert torch.allclose(y.float().cpu(), want, atol=tol(dt, 2e-5, 1.5e-2), rtol=tol(dt, 1e-5, 1e-2)), float((y.float().cpu() - want).abs().max())
    # round 6: the walk order of a frame's queries (moy_query_order: Morton order of the reference points' level-0 cells) changes no
    # output bit; the order itself is a permutation of every frame, sorted by cell code with ties in query order
    perm = ops.query_order(ref_box.to(DEV), B, Lq, H0, W0)
    yq = ops.msda_raw0(xbuf[:, :128], wc.to(DEV, dt).contiguous(), bc.to(DEV), planes, B, shapes, offaw.to(DEV), ref_box.to(DEV), Lq, perm=perm)
    gq = torch.Generator().manual_seed(11)
    rperm = torch.stack([torch.randperm(Lq, generator=gq) for _ in range(B)]).to(DEV, torch.int32)
    yr = ops.msda_raw0(xbuf[:, :128], wc.to(DEV, dt).contiguous(), bc.to(DEV), planes, B, shapes, offaw.to(DEV), ref_box.to(DEV), Lq, perm=rperm)
    torch.cuda.synchronize()
    assert torch.equal(yq, y) and torch.equal(yr, y)
    pc = perm.cpu().long()
    assert all(torch.equal(pc[b].sort().values, torch.arange(Lq)) for b in range(B))
    def _code(v):
        v = (v | (v << 4)) & 0x0f0f; v = (v | (v << 2)) & 0x3333; v = (v | (v << 1)) & 0x5555
        return v
    rbx = ref_box.view(B, Lq, 4)
    xi = (rbx[..., 0] * W0).int().clamp(0, min(W0, 256) - 1).long(); yi = (rbx[..., 1] * H0).int().clamp(0, min(H0, 256) - 1).long()
    keys = ((_code(xi) | (_code(yi) << 1)) << 10) | torch.arange(Lq)[None]
    assert torch.equal(pc, keys.argsort(1))
    # round 6: the fp32 form (the fp32 engines' folded head): fp32 tensors, fp32 sums, exact fp32 matrix instruction -- against the
    # same oracle on the same (16-bit representable) operands: 2e-5; a packed weight matrix or a short head stride is refused
    xf = torch.zeros(B * H0 * W0, ld0, device=DEV, dtype=torch.float32)
    xf[:, :128] = x.to(DEV)
    pf = planes.float().contiguous() if planes is not None else None
    yf = ops.msda_raw0(xf[:, :128], wc.to(DEV).contiguous(), bc.to(DEV), pf, B, shapes, offaw.to(DEV), ref_box.to(DEV), Lq)
    yfq = ops.msda_raw0(xf[:, :128], wc.to(DEV).contiguous(), bc.to(DEV), pf, B, shapes, offaw.to(DEV), ref_box.to(DEV), Lq, perm=rperm)
    torch.cuda.synchronize()
    assert yf.dtype == torch.float32 and torch.equal(yfq, yf)
    assert torch.allclose(yf.cpu(), want, atol=2e-5, rtol=1e-5), float((yf.cpu() - want).abs().max())
    with pytest.raises(L.MoyoloError):
        ops.msda_raw0(xf[:, :128], wc.to(DEV).contiguous(), bc.to(DEV), pf, B, shapes, offaw.to(DEV), ref_box.to(DEV), Lq, packed=True)
    if planes is not None and B > 1:
        with pytest.raises(L.MoyoloError):      # head planes shorter than B * S1 tokens
            ops.msda_raw0(xbuf[:, :128], wc.to(DEV, dt).contiguous(), bc.to(DEV), planes, B, shapes, offaw.to(DEV), ref_box.to(DEV), Lq,
                          head_stride=(B - 1) * S1 * 32)


@pytest.mark.parametrize("dt", DT)
def test_ms_deform_attn_forward_kats(dt):
    """The reference operator API on its own KAT construction (MOTR/models/ops/test.py:21-30),
    expected values produced by the reference's torch op in the build container."""
    g = golden("msda_kat")
    for name in ("kat_tiny", "kat_heads8", "kat_odd"):
        v, loc, aw = (torch.from_numpy(g[f"{name}.{k}"]) for k in ("value", "loc", "aw"))
        shapes = torch.from_numpy(g[name + ".shapes"])
        lsi = torch.cat((shapes.new_zeros(1), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]))
        y = ops.ms_deform_attn_forward(v.to(DEV, dt), shapes.to(DEV), lsi.to(DEV), loc.to(DEV, dt), aw.to(DEV, dt), 64)
        want = torch.from_numpy(g[name + ".out"])
        if dt == torch.float32:
            assert torch.allclose(y.cpu(), want, atol=1e-3, rtol=1e-2)       # the reference's own fp32 bar (ops/test.py:50)
            assert torch.allclose(y.cpu(), want, atol=1e-7, rtol=1e-5)       # ours
        elif dt == torch.bfloat16:
            assert torch.allclose(y.float().cpu(), want, atol=3e-4, rtol=3e-2)
        else:                                                               # fp16 (config C5): 11-bit operands, fp32 accumulation
            assert torch.allclose(y.float().cpu(), want, atol=4e-5, rtol=4e-3)


def _kat(g, name, dt):
    v, loc, aw = (torch.from_numpy(g[f"{name}.{k}"]).to(dt) for k in ("value", "loc", "aw"))
    shapes = torch.from_numpy(g[name + ".shapes"])
    lsi = torch.cat((shapes.new_zeros(1), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]))
    return v, loc, aw, shapes, lsi


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
def test_ms_deform_attn_backward_kats(dt):
    """ms_deform_attn_backward (ms_deform_attn.h:42-62) vs autograd through the reference's torch op, fixtures from
    tests/golden/make_golden.py; fp64 is the precision the reference's own gradcheck runs in (ops/test.py:66-86)."""
    g = golden("msda_kat")
    tag, atol = ("", 2e-7) if dt == torch.float32 else ("_f64", 1e-14)
    for name in ("kat_tiny", "kat_heads8", "kat_odd"):
        v, loc, aw, shapes, lsi = _kat(g, name, dt)
        go = torch.from_numpy(g[name + ".gout"]).to(dt)
        if dt == torch.float64:     # fp64 forward of the operator API too
            y = ops.ms_deform_attn_forward(v.to(DEV), shapes.to(DEV), lsi.to(DEV), loc.to(DEV), aw.to(DEV), 64)
            assert torch.allclose(y.cpu(), torch.from_numpy(g[name + ".out_f64"]), atol=1e-14, rtol=1e-10)
        got = ops.ms_deform_attn_backward(v.to(DEV), shapes.to(DEV), lsi.to(DEV), loc.to(DEV), aw.to(DEV), go.to(DEV), 64)
        for k, t in zip(("gvalue", "gloc", "gaw"), got):
            assert torch.allclose(t.cpu(), torch.from_numpy(g[f"{name}.{k}{tag}"]), atol=atol, rtol=1e-5), (name, k)


@pytest.mark.parametrize("D", [30, 32, 64, 71, 1025, 2048, 3096])
def test_ms_deform_attn_backward_channels(D):
    """ALL the channel counts the reference's gradient test sweeps (MOTR/models/ops/test.py:85-86), against the
    oracle's analytic backward in fp64, through the autograd Function (functions/ms_deform_attn_func.py:24-41)."""
    from mo_yolo_amd.modules import MSDeformAttnFunction
    N, M, Lq, P = 2, 2, 5, 3
    shapes_l = [(6, 4), (3, 2)]
    S = sum(h * w for h, w in shapes_l)
    gen = torch.Generator().manual_seed(D)
    value = torch.rand(N, S, M, D, generator=gen, dtype=torch.float64) * 0.01
    loc = torch.rand(N, Lq, M, 2, P, 2, generator=gen, dtype=torch.float64) * 1.3 - 0.15
    aw = torch.rand(N, Lq, M, 2, P, generator=gen, dtype=torch.float64) + 1e-5
    aw = aw / aw.sum(-1, keepdim=True).sum(-2, keepdim=True)
    go = torch.rand(N, Lq, M * D, generator=gen, dtype=torch.float64) - 0.5
    shapes = torch.tensor(shapes_l, dtype=torch.int64)
    lsi = torch.cat((shapes.new_zeros(1), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]))
    v_, l_, a_ = (t.to(DEV).requires_grad_(True) for t in (value, loc, aw))
    y = MSDeformAttnFunction.apply(v_, shapes.to(DEV), lsi.to(DEV), l_, a_, 2)
    assert torch.allclose(y.detach().cpu(), O.msda_core(value, shapes_l, loc, aw), atol=1e-14, rtol=1e-10)
    y.backward(go.to(DEV))
    for t, w in zip((v_.grad, l_.grad, a_.grad), O.msda_core_backward(value, shapes_l, loc, aw, go)):
        assert torch.allclose(t.cpu(), w, atol=1e-13, rtol=1e-9)


def test_ms_deform_attn_errors():
    """Error behaviour of the operator: CPU tensors and batch % im2col_step (ms_deform_attn.h:39, ms_deform_attn_cuda.cu:49)."""
    g = golden("msda_kat")
    v, loc, aw, shapes, lsi = _kat(g, "kat_heads8", torch.float32)
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        ops.ms_deform_attn_forward(v, shapes, lsi, loc, aw, 64)
    v3 = torch.cat((v, v[:1])).to(DEV)
    with pytest.raises(RuntimeError, match="must divide"):
        ops.ms_deform_attn_forward(v3, shapes.to(DEV), lsi.to(DEV), torch.cat((loc, loc[:1])).to(DEV), torch.cat((aw, aw[:1])).to(DEV), 2)


@pytest.mark.parametrize("src_hw,dst_hw", [((1080, 1920), (608, 1088)), ((480, 640), (608, 1088)), ((1216, 2176), (608, 1088)),
                                           ((37, 53), (64, 96)), ((64, 96), (64, 96)), ((3, 2), (32, 32)), ((700, 1), (32, 64))])
def test_resize_linear_u8_bit_exact(src_hw, dst_hw):
    """moy_resize_linear_u8 vs the oracle's restatement of cv2.resize INTER_LINEAR (LetterBox scaleFill, augment.py:573-576):
    byte work, bit-exact; shrink, enlarge, the 2x INTER_AREA reroute, unit scale, degenerate 1-pixel-wide sources."""
    from oracle.preprocess_oracle import resize_linear_u8 as ref
    B = 2
    src = np.random.default_rng(src_hw[0]).integers(0, 256, (B, *src_hw, 3), dtype=np.uint8)
    got = ops.resize_linear_u8(torch.from_numpy(src).to(DEV), dst_hw).cpu().numpy()
    for b in range(B):
        assert np.array_equal(got[b], ref(src[b], dst_hw)), b
    # a pitched source (a crop of a wider frame): explicit row / image strides
    if src_hw[1] > 8:
        crop = torch.from_numpy(src).to(DEV)[:, :, 4:]
        got = ops.resize_linear_u8(crop, dst_hw).cpu().numpy()
        assert np.array_equal(got[1], ref(np.ascontiguousarray(src[1, :, 4:]), dst_hw))


def test_assign_post_semantics():
    B, nq, nc = 4, 300, 2
    logits = rnd(B, nq, nc, seed=1, scale=4.0)
    logits[1] = -9.0                                          # nothing active -> detection fallback
    logits[2] = 9.0                                           # all 300 active
    logits[3, :, :] = torch.logit(torch.tensor(0.3))          # between conf 0.25 and birth 0.4: fallback rows
    boxes = torch.rand(B, nq, 4, generator=torch.Generator().manual_seed(2))
    out = ops.assign_post(logits.to(DEV), boxes.to(DEV), 0.4, 0.25, (1088.0, 608.0))
    out = {k: v.cpu() for k, v in out.items()}
    for b in range(B):
        y = torch.cat((boxes[b], logits[b].sigmoid()), -1)
        assert torch.allclose(out["y"][b], y, atol=1e-6)
        scores = logits[b].sigmoid().max(-1).values
        ids = O.assign_ids(scores)
        assert torch.equal(out["obj_idxes"][b], ids), b
        rows, tid = O.postprocess(y, logits[b], ids, 0.25, orig_hw=(608, 1088))
        n = int(out["n_rows"][b])
        assert n == rows.shape[0]
        assert torch.allclose(out["rows"][b, :n], rows, atol=1e-3, rtol=1e-6)
        if tid is None:
            assert int(out["n_ids"][b]) == -1
        else:
            k = int(out["n_ids"][b])
            assert k == tid.numel() and torch.equal(out["track_id"][b, :k], tid)


@pytest.mark.parametrize("dt", DT)
def test_small_helpers(dt):
    x = q(rnd(100, 64, seed=1), dt)
    rows = torch.tensor([5, 5, 99, 0], dtype=torch.int32)
    assert torch.equal(ops.gather_rows(x.to(DEV, dt), rows.to(DEV)).float().cpu(), x[rows.long()])
    f = rnd(10, 8, seed=2)
    assert torch.equal(ops.cast_f32_to(f.to(DEV), dt).cpu(), f.to(dt))
    assert torch.allclose(ops.sigmoid_f32(f.to(DEV)).cpu(), f.sigmoid(), atol=1e-6)

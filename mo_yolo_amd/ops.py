"""Functional tensor-level wrappers over the C ABI (shape checks on the host, outputs allocated
with torch).  These are what the drop-in module classes (mo_yolo_amd.modules) and the parity tests
call; the TrackEngine pre-builds the same calls into a static plan instead."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _code(t: torch.Tensor):
    if t.dtype == torch.float32:
        return L.F32
    if t.dtype == torch.bfloat16:
        return L.BF16
    if t.dtype == torch.float16:
        return L.F16
    raise TypeError(f"unsupported dtype {t.dtype}")


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            # same convention as the reference op: ms_deform_attn.h:39 AT_ERROR("Not implemented on the CPU")
            raise RuntimeError("Not implemented on the CPU")


def _ld(t):
    assert t.dim() == 2 and t.stride(1) == 1, "row-major 2-D view expected"
    return t.stride(0)


def pad_weight(w2d: torch.Tensor, dtype) -> torch.Tensor:
    """[N, K] -> [N, Kpad] (Kpad multiple of 64 for bf16 / 32 for f32), zero padded, in `dtype`."""
    bk = 32 if dtype == torch.float32 else 64
    N, K = w2d.shape
    out = torch.zeros(N, (K + bk - 1) // bk * bk, device=w2d.device, dtype=torch.float32)
    out[:, :K] = w2d.float()
    return out.to(dtype).contiguous()


def split_weight(w2d: torch.Tensor) -> torch.Tensor:
    """[N, K] fp32 -> fp16 [2, N, Kpad32]: the operand form of MOY_F32X3 (moyolo.h): plane 0 = fp16(w), plane 1 = fp16((w - plane 0) * 2^11)."""
    w = pad_weight(w2d, torch.float32)
    hi = w.to(torch.float16)
    lo = ((w - hi.float()) * 2048.0).to(torch.float16)
    return torch.stack([hi, lo]).contiguous()


def gemm(A, Wp, N, K, *, out=None, ksize=1, stride=1, geom=None, scale=None, shift=None, act=L.ACT_NONE, A2=None,
         a_rows=None, a_mask=None, mask_period=0, R=None, ln=None, out_f32=False, M=None, c_rpb=0, c_bstride=0,
         dot=None, store=True, pre=None, a2_cols=0, planes=None, runs=None, dot_out=None, post=None, split_f16=False):
    """See moy_gemm.  store=False (with dot): C = NULL, only the fused head's output is produced (returned as (None, dot_out)).
    A: 2-D row-major view [rows, >=Cin] (channels-last pixels or tokens)."""
    _need_gpu(A, Wp)
    a = L.GemmArgs()
    a.A, a.lda = A.data_ptr(), _ld(A)
    if A2 is not None:
        assert _ld(A2) == _ld(A)
        a.A2 = A2.data_ptr()
    if a_rows is not None:
        assert a_rows.dtype == torch.int32
        a.a_rows, a.a_rows_bound = a_rows.data_ptr(), A.shape[0]
    if a_mask is not None:
        assert a_mask.dtype == torch.uint8
        a.a_mask, a.mask_period = a_mask.data_ptr(), mask_period
    if ksize == 3:
        B, Hin, Win, Hout, Wout, Cin = geom
        a.B, a.Hin, a.Win, a.Hout, a.Wout, a.Cin = geom
        M = B * Hout * Wout
    elif M is None:
        M = a_rows.numel() if a_rows is not None else A.shape[0]
    a.W, a.M, a.N, a.K, a.ksize, a.stride = Wp.data_ptr(), M, N, K, ksize, stride
    if scale is not None:
        a.scale = scale.data_ptr()
    if shift is not None:
        a.shift = shift.data_ptr()
    a.act = act
    if R is not None:
        a.R, a.ldr = R.data_ptr(), _ld(R)
    if ln is not None:
        a.ln_g, a.ln_b = ln[0].data_ptr(), ln[1].data_ptr()
    if not store:
        assert dot is not None and out is None
        a.out_f32, a.dtype = int(out_f32), _code(A)
    else:
        if out is None:
            rows = M if not c_rpb else (M // c_rpb) * c_bstride
            out = torch.empty(rows, N if post is None else post[0].shape[0], device=A.device, dtype=torch.float32 if out_f32 else A.dtype)
        a.C, a.ldc, a.out_f32, a.dtype = out.data_ptr(), _ld(out), int(out_f32), _code(A)
    a.c_rows_per_batch, a.c_batch_stride = c_rpb, c_bstride
    if pre is not None:     # (fp32 [B*(H/2)*(W/2), >=N], H, W): accumulator seed = nearest-2x upsampled half-resolution product
        pt, ph, pw = pre
        assert pt.dtype == torch.float32
        a.pre, a.ld_pre, a.pre_h, a.pre_w = pt.data_ptr(), _ld(pt), ph, pw
    a.a2_cols = a2_cols
    if planes is not None:   # (plane_cols, plane_stride in elements): `out` is the first plane [M, plane_cols]
        a.plane_cols, a.plane_stride = planes
    if runs is not None:     # score pass over row runs: dict(period, levels=[(tok0, pitch, len, rows), ...], a_period=0, a_off=0)
        a.run_levels, a.run_period = len(runs["levels"]), runs["period"]
        for i, (t0, pit, ln_, rw) in enumerate(runs["levels"]):
            a.run_tok0[i], a.run_pitch[i], a.run_len[i], a.run_rows[i] = t0, pit, ln_, rw
        a.run_a_period, a.run_a_off = runs.get("a_period", 0), runs.get("a_off", 0)
    if post is not None:    # (W2 padded [n2, ceil64(N)], scale2, shift2, act2): a 1x1 conv on the finished tiles; `out` has n2 columns
        pw_, ps_, ph_, pact = post
        a.post_W, a.post_n, a.post_act = pw_.data_ptr(), pw_.shape[0], pact
        a.post_scale = ps_.data_ptr() if ps_ is not None else None
        a.post_shift = ph_.data_ptr() if ph_ is not None else None
    if dot is not None:     # (w fp32 [n, 256], b fp32 [n]) fused behind the LayerNorm
        dw, db = dot
        if dot_out is None:
            dot_out = torch.empty(M, dw.shape[0], device=A.device, dtype=torch.float32)
        a.dot_w, a.dot_b, a.dot_out, a.dot_n = dw.data_ptr(), db.data_ptr(), dot_out.data_ptr(), dw.shape[0]
    if split_f16:           # MOY_F32X3: fp32 tensors, split-fp16 matrix arithmetic; W pre-split (split_weight)
        assert A.dtype == torch.float32 and Wp.dtype == torch.float16 and Wp.dim() == 3 and Wp.shape[0] == 2 and Wp.shape[1] == N
        a.dtype = L.F32X3
    L.check(L.lib().moy_gemm(C.byref(a), _st()), "moy_gemm")
    return out if dot is None else (out, dot_out)


def stem_conv(x, w27, scale, shift, dtype):
    _need_gpu(x)
    if x.dtype == torch.uint8:
        B, H, W, _ = x.shape
        fmt = 0
    else:
        B, _, H, W = x.shape
        fmt = 1
    cout = w27.shape[1]
    out = torch.empty(B * (H // 2) * (W // 2), cout, device=x.device, dtype=dtype)
    L.check(L.lib().moy_stem_conv(x.data_ptr(), fmt, B, H, W, w27.data_ptr(), scale.data_ptr(), shift.data_ptr(), cout,
                                  out.data_ptr(), cout, _code(out), _st()), "moy_stem_conv")
    return out


def stem_weights_mfma(w):
    """[Cout, 3, 3, 3] fp32 conv weight -> bf16 [Cout, 32], k = (ky*3+kx)*3 + c (zero padded from 27)."""
    cout = w.shape[0]
    out = torch.zeros(cout, 32, device=w.device, dtype=torch.float32)
    out[:, :27] = w.permute(0, 2, 3, 1).reshape(cout, 27)
    return out.to(torch.bfloat16).contiguous()


def stem_weights_x3(w):
    """[Cout, 3, 3, 3] fp32 conv weight (c_rgb) -> fp16 [2, Cout, 32] for moy_stem_conv_x3: moy_stem_l1_fused's k order (slice q < 3 = tap
    row ky = q over the 8 bytes (kx, c_bgr) of a window row, slice 3 = the ninth byte (kx 2, red) of the three rows, then zeros);
    plane 0 = fp16(w), plane 1 = fp16((w - plane 0) * 2^11) (the MOY_F32X3 operand form, split_weight)."""
    cout = w.shape[0]
    w = w.float()
    w32 = torch.zeros(cout, 32, device=w.device, dtype=torch.float32)
    for qq in range(3):
        for e in range(8):
            w32[:, qq * 8 + e] = w[:, 2 - e % 3, qq, e // 3]
    for e in range(3):
        w32[:, 24 + e] = w[:, 0, e, 2]
    hi = w32.to(torch.float16)
    lo = ((w32 - hi.float()) * 2048.0).to(torch.float16)
    return torch.stack([hi, lo]).contiguous()


def stem_weights_fused(w, dtype=None):
    """[32, 3, 3, 3] fp32 stem weight (c_rgb) -> IEEE half [32, 32] in moy_stem_l1_fused's k order: slice q < 3 = tap row ky = q over
    the 8 bytes (kx, c_bgr) of a window row, slice 3 = the ninth byte (kx 2, c_bgr 2 = red) of the three rows.  Halfs for BOTH engine
    types: the kernel feeds the frame bytes to the matrix cores as the halfs 1024 + x (`dtype` is accepted for old call sites)."""
    assert tuple(w.shape) == (32, 3, 3, 3)
    out = torch.zeros(32, 32, device=w.device, dtype=torch.float32)
    for qq in range(3):
        for e in range(8):
            out[:, qq * 8 + e] = w[:, 2 - e % 3, qq, e // 3]
    for e in range(3):
        out[:, 24 + e] = w[:, 0, e, 2]
    return out.to(torch.float16).contiguous()


def stem_l1_fused(x_u8, w0, scale0, shift0, w1pad, scale1, shift1, dtype, out=None):
    """Layers 0 + 1 (both 3x3 stride 2 + BN + SiLU) of the backbone from uint8 BGR frames in one launch; out [B*(H/4)*(W/4), 64]."""
    _need_gpu(x_u8)
    B, H, W, _ = x_u8.shape
    if out is None:
        out = torch.empty(B * (H // 4) * (W // 4), 64, device=x_u8.device, dtype=dtype)
    L.check(L.lib().moy_stem_l1_fused(x_u8.data_ptr(), B, H, W, w0.data_ptr(), scale0.data_ptr(), shift0.data_ptr(), w1pad.data_ptr(),
                                      scale1.data_ptr(), shift1.data_ptr(), out.data_ptr(), _ld(out), _code(out), _st()),
            "moy_stem_l1_fused")
    return out


def c2f_fused(x, B, H, W, cv1, m1, m2, cv2, out=None):
    """C2f block 64 -> [32 | 32] -> 64, n = 1, shortcut, in one launch (csrc/c2f_fused.hip).  x [B*H*W, 64] (a channel slice of a
    wider buffer is fine); cv1 / m1 / m2 / cv2 = (padded weight [N, Kpad], scale fp32 [N], shift fp32 [N]) with the weights of the
    3x3 convs as [32, (ky*3+kx)*32 + c]; out [B*H*W, 64]."""
    _need_gpu(x)
    if out is None:
        out = torch.empty(B * H * W, 64, device=x.device, dtype=x.dtype)
    a = L.C2fArgs()
    a.x, a.ldx, a.B, a.H, a.W = x.data_ptr(), _ld(x), B, H, W
    a.w_cv1, a.kp_cv1, a.scale_cv1, a.shift_cv1 = cv1[0].data_ptr(), cv1[0].shape[1], cv1[1].data_ptr(), cv1[2].data_ptr()
    a.w_m1, a.scale_m1, a.shift_m1 = m1[0].data_ptr(), m1[1].data_ptr(), m1[2].data_ptr()
    a.w_m2, a.scale_m2, a.shift_m2, a.kp_m = m2[0].data_ptr(), m2[1].data_ptr(), m2[2].data_ptr(), m1[0].shape[1]
    assert m1[0].shape[1] == m2[0].shape[1]
    a.w_cv2, a.kp_cv2, a.scale_cv2, a.shift_cv2 = cv2[0].data_ptr(), cv2[0].shape[1], cv2[1].data_ptr(), cv2[2].data_ptr()
    a.out, a.ldo, a.dtype = out.data_ptr(), _ld(out), _code(out)
    L.check(L.lib().moy_c2f_fused(C.byref(a), _st()), "moy_c2f_fused")
    return out


def stem_conv_mfma(x_u8, wpad, scale, shift):
    _need_gpu(x_u8)
    B, H, W, _ = x_u8.shape
    cout = wpad.shape[0]
    out = torch.empty(B * (H // 2) * (W // 2), cout, device=x_u8.device, dtype=torch.bfloat16)
    L.check(L.lib().moy_stem_conv_mfma(x_u8.data_ptr(), B, H, W, wpad.data_ptr(), scale.data_ptr(), shift.data_ptr(), cout,
                                       out.data_ptr(), cout, _st()), "moy_stem_conv_mfma")
    return out


def stem_conv_x3(x_u8, wsplit, scale, shift):
    """Stem of the split-fp16 engine: uint8 [B, H, W, 3] BGR -> fp32 [B*(H/2)*(W/2), Cout] (moy_stem_conv_x3)."""
    _need_gpu(x_u8)
    B, H, W, _ = x_u8.shape
    cout = wsplit.shape[1]
    out = torch.empty(B * (H // 2) * (W // 2), cout, device=x_u8.device, dtype=torch.float32)
    L.check(L.lib().moy_stem_conv_x3(x_u8.data_ptr(), B, H, W, wsplit.data_ptr(), scale.data_ptr(), shift.data_ptr(), cout,
                                     out.data_ptr(), cout, _st()), "moy_stem_conv_x3")
    return out


def sppf_pool(x, B, H, W):
    _need_gpu(x)
    Cc = x.shape[1]
    ys = [torch.empty(B * H * W, Cc, device=x.device, dtype=x.dtype) for _ in range(3)]
    L.check(L.lib().moy_sppf_pool(x.data_ptr(), _ld(x), B, H, W, Cc, ys[0].data_ptr(), ys[1].data_ptr(), ys[2].data_ptr(),
                                  Cc, _code(x), _st()), "moy_sppf_pool")
    return ys


def upsample2x(x, B, H, W):
    _need_gpu(x)
    Cc = x.shape[1]
    y = torch.empty(B * 4 * H * W, Cc, device=x.device, dtype=x.dtype)
    L.check(L.lib().moy_upsample2x(x.data_ptr(), _ld(x), B, H, W, Cc, y.data_ptr(), Cc, _code(x), _st()), "moy_upsample2x")
    return y


def rowdot(X, Wt, bias, mode=0, aux=None, aux_rows=None, x_rows=None):
    _need_gpu(X)
    N, K = Wt.shape
    M = x_rows.numel() if x_rows is not None else X.shape[0]
    y = torch.empty(M, N, device=X.device, dtype=torch.float32)
    L.check(L.lib().moy_rowdot(X.data_ptr(), _ld(X), x_rows.data_ptr() if x_rows is not None else None, M, K,
                               Wt.data_ptr(), bias.data_ptr() if bias is not None else None, N, mode,
                               aux.data_ptr() if aux is not None else None,
                               aux_rows.data_ptr() if aux_rows is not None else None, y.data_ptr(), _code(X), _st()),
            "moy_rowdot")
    return y


def mlp_head(X, W0p, b0, W1p, b1, w2, b2, mode=0, aux=None, aux_rows=None, x_rows=None):
    """See moy_mlp_head: X [rows, 256] 16-bit, W0p / W1p padded [256, 256] in X's dtype, b0 / b1 / w2 [4, 256] / b2 fp32."""
    _need_gpu(X)
    M = x_rows.numel() if x_rows is not None else X.shape[0]
    y = torch.empty(M, 4, device=X.device, dtype=torch.float32)
    L.check(L.lib().moy_mlp_head(X.data_ptr(), _ld(X), x_rows.data_ptr() if x_rows is not None else None, M, W0p.data_ptr(),
                                 b0.data_ptr(), W1p.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), mode,
                                 aux.data_ptr() if aux is not None else None,
                                 aux_rows.data_ptr() if aux_rows is not None else None, y.data_ptr(), _code(X), _st()),
            "moy_mlp_head")
    return y


def pack_mfma_a(W):
    """Row-major 16-bit W [N, K] (N, K multiples of 32) -> the same elements in MFMA-fragment order (include/moyolo.h, above
    moy_decoder_tail_args): blocks [N/32][K/32] of 2 KB = [j][lane = q*16 + r][8 elements] with row 32 g + 16 j + r, column 32 pn + 8 q + e."""
    N, K = W.shape
    assert N % 32 == 0 and K % 32 == 0 and W.element_size() == 2
    return W.contiguous().view(N // 32, 2, 16, K // 32, 4, 8).permute(0, 3, 1, 4, 2, 5).contiguous().view(N, K)


def decoder_tail(samp, e1, Wp, bp, ln2, W1, b1, W2, b2, ln3, B0, c0, B1, c1, w2, c2, ref_in, packed=False, next_qkv=None):
    """See moy_decoder_tail.  samp / e1 [M, 256] 16-bit; weights in that dtype ([out, in]; packed=True: each through pack_mfma_a);
    vectors fp32; ref_in fp32 [M, 4].  Returns (out [M, 256], ref_out fp32 [M, 4]); with next_qkv = (Wqkv [768, 256], bqkv fp32 [768],
    qpos [M, 256]) also the next layer's q | k | v [M, 768]."""
    _need_gpu(samp, e1)
    M = samp.shape[0]
    out = torch.empty(M, 256, device=samp.device, dtype=samp.dtype)
    ref_out = torch.empty(M, 4, device=samp.device, dtype=torch.float32)
    t = L.DecoderTailArgs()
    t.samp, t.ld_samp, t.e1, t.ld_e1, t.M = samp.data_ptr(), _ld(samp), e1.data_ptr(), _ld(e1), M
    t.Wp, t.bp, t.ln2_g, t.ln2_b = Wp.data_ptr(), bp.data_ptr(), ln2[0].data_ptr(), ln2[1].data_ptr()
    t.W1, t.b1, t.W2, t.b2, t.d_ffn = W1.data_ptr(), b1.data_ptr(), W2.data_ptr(), b2.data_ptr(), W1.shape[0]
    t.ln3_g, t.ln3_b, t.out, t.ld_out = ln3[0].data_ptr(), ln3[1].data_ptr(), out.data_ptr(), 256
    t.B0, t.c0, t.B1, t.c1, t.w2, t.c2 = B0.data_ptr(), c0.data_ptr(), B1.data_ptr(), c1.data_ptr(), w2.data_ptr(), c2.data_ptr()
    t.ref_in, t.ref_out, t.dtype, t.w_packed = ref_in.data_ptr(), ref_out.data_ptr(), _code(samp), int(packed)
    qkv = None
    if next_qkv is not None:
        Wqkv, bqkv, qpos = next_qkv
        qkv = torch.empty(M, 768, device=samp.device, dtype=samp.dtype)
        t.Wqkv, t.bqkv, t.qkv, t.ld_qkv, t.qpos, t.ld_qpos = Wqkv.data_ptr(), bqkv.data_ptr(), qkv.data_ptr(), 768, qpos.data_ptr(), _ld(qpos)
    L.check(L.lib().moy_decoder_tail(C.byref(t), _st()), "moy_decoder_tail")
    return (out, ref_out) if qkv is None else (out, ref_out, qkv)


def decoder_mid(attn, x, qpos, Wo, bo, ln1, Woa, boa, n_oa, packed=False):
    """See moy_decoder_mid.  attn / x / qpos [M, 256] 16-bit; Wo [256, 256], Woa [max(256, n_oa), 256] in that dtype (zero rows past
    n_oa); vectors fp32.  Returns (e1 [M, 256], offaw fp32 [M, n_oa])."""
    _need_gpu(attn, x, qpos)
    M = attn.shape[0]
    e1 = torch.empty(M, 256, device=attn.device, dtype=attn.dtype)
    offaw = torch.empty(M, n_oa, device=attn.device, dtype=torch.float32)
    t = L.DecoderMidArgs()
    t.attn, t.ld_attn, t.x, t.ld_x, t.qpos, t.ld_qpos, t.M = attn.data_ptr(), _ld(attn), x.data_ptr(), _ld(x), qpos.data_ptr(), _ld(qpos), M
    t.Wo, t.bo, t.ln_g, t.ln_b = Wo.data_ptr(), bo.data_ptr(), ln1[0].data_ptr(), ln1[1].data_ptr()
    t.Woa, t.boa, t.n_oa = Woa.data_ptr(), boa.data_ptr(), n_oa
    t.e1, t.ld_e1, t.offaw, t.ld_oa, t.dtype, t.w_packed = e1.data_ptr(), 256, offaw.data_ptr(), n_oa, _code(attn), int(packed)
    L.check(L.lib().moy_decoder_mid(C.byref(t), _st()), "moy_decoder_mid")
    return e1, offaw


def topk(scores, nq, valid=None):
    """scores fp32 [B, S, nc] -> (idx_local int32 [B, nq], idx_global, n_masked int32 [B])."""
    _need_gpu(scores)
    B, S, nc = scores.shape
    il = torch.empty(B, nq, device=scores.device, dtype=torch.int32)
    ig = torch.empty_like(il)
    nm = torch.zeros(B, device=scores.device, dtype=torch.int32)
    L.check(L.lib().moy_topk(scores.data_ptr(), B, S, nc, nq, valid.data_ptr() if valid is not None else None,
                             il.data_ptr(), ig.data_ptr(), nm.data_ptr(), _st()), "moy_topk")
    return il, ig, nm


def pos2posemb(pos, dtype):
    _need_gpu(pos)
    M = pos.shape[0]
    out = torch.empty(M, 256, device=pos.device, dtype=dtype)
    L.check(L.lib().moy_pos2posemb(pos.data_ptr(), M, out.data_ptr(), 256, _code(out), _st()), "moy_pos2posemb")
    return out


def mha_core(qkv, B, Lq, nh):
    _need_gpu(qkv)
    E = qkv.shape[1] // 3
    out = torch.empty(B * Lq, E, device=qkv.device, dtype=qkv.dtype)
    L.check(L.lib().moy_mha_core(qkv.data_ptr(), _ld(qkv), B, Lq, nh, E, out.data_ptr(), E, _code(qkv), _st()), "moy_mha_core")
    return out


def msda_fused(value, B, S, shapes, offaw, ref, Lq, head_planes=False):
    """value: [B*S, 256] (head-major channels), or with head_planes [8, B*S, 32] contiguous."""
    _need_gpu(value, offaw, ref)
    nl = len(shapes)
    sh = (C.c_int32 * (2 * nl))(*[int(v) for hw in shapes for v in hw])
    out = torch.empty(B * Lq, 256, device=value.device, dtype=value.dtype)
    ldv, hs = (32, B * S * 32) if head_planes else (_ld(value), 32)
    if head_planes:
        assert value.is_contiguous() and tuple(value.shape) == (8, B * S, 32)
    L.check(L.lib().moy_msda_fused(value.data_ptr(), ldv, hs, B, S, sh, nl, offaw.data_ptr(), _ld(offaw), ref.data_ptr(),
                                   Lq, out.data_ptr(), 256, _code(value), _st()), "moy_msda_fused")
    return out


def query_order(ref, B, Lq, H0, W0):
    """moy_query_order: perm int32 [B, Lq] = every frame's queries sorted by the Morton code of the (H0 x W0)-grid cell of their
    reference box centre (ref fp32 [B*Lq, 4])."""
    _need_gpu(ref)
    assert ref.dtype == torch.float32 and ref.is_contiguous() and ref.numel() == B * Lq * 4
    perm = torch.empty(B, Lq, device=ref.device, dtype=torch.int32)
    L.check(L.lib().moy_query_order(ref.data_ptr(), B, Lq, H0, W0, perm.data_ptr(), _st()), "moy_query_order")
    return perm


def msda_raw0(x0, wc, bc, planes, B, shapes, offaw, ref, Lq, packed=False, head_stride=None, perm=None):
    """moy_msda_raw0: level 0 gathered raw from x0 [B*H0*W0, >= 128] (channel-slice view) and projected with wc [256, 128] / bc [256]
    after the bilinear sum; levels 1.. from head planes [8, B*S1, 32] (None when there is one level)."""
    _need_gpu(x0, wc, bc, offaw, ref)
    nl = len(shapes)
    a = L.MsdaRawArgs()
    sh = (C.c_int32 * (2 * nl))(*[int(v) for hw in shapes for v in hw])
    S1 = sum(h * w for h, w in shapes[1:])
    out = torch.empty(B * Lq, 256, device=x0.device, dtype=x0.dtype)
    a.x0, a.ld0, a.wc, a.bc = x0.data_ptr(), _ld(x0), wc.data_ptr(), bc.data_ptr()
    if planes is not None:
        assert planes.is_contiguous() and tuple(planes.shape) == (8, B * S1, 32) and planes.dtype == x0.dtype
        a.planes, a.head_stride = planes.data_ptr(), (B * S1 * 32 if head_stride is None else head_stride)
    a.S1, a.B, a.Lq, a.L, a.shapes_hw = S1, B, Lq, nl, C.cast(sh, C.c_void_p)
    a.offaw, a.ld_oa, a.ref, a.out, a.ldo, a.dtype = offaw.data_ptr(), _ld(offaw), ref.data_ptr(), out.data_ptr(), 256, _code(x0)
    a.wc_packed = int(packed)          # wc through pack_mfma_a
    if perm is not None:               # the order in which the kernel walks a frame's queries (moy_query_order); outputs unchanged
        assert perm.dtype == torch.int32 and perm.is_contiguous() and tuple(perm.shape) == (B, Lq)
        a.perm = perm.data_ptr()
    assert wc.is_contiguous() and tuple(wc.shape) == (256, 128) and wc.dtype == x0.dtype and bc.dtype == torch.float32
    L.check(L.lib().moy_msda_raw0(C.byref(a), _st()), "moy_msda_raw0")
    return out


def _msda_check(tensors, value, im2col_step):
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError("Not implemented on the CPU")      # ms_deform_attn.h:39,61
        if not t.is_contiguous():
            raise RuntimeError("tensor has to be contiguous")     # ms_deform_attn_cuda.cu:28-32,91-96
    step = min(value.shape[0], int(im2col_step))
    if value.shape[0] % step:
        raise RuntimeError(f"batch({value.shape[0]}) must divide im2col_step({step})")   # ms_deform_attn_cuda.cu:49,118


def ms_deform_attn_forward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, im2col_step=64):
    """Same call as MultiScaleDeformableAttention.ms_deform_attn_forward (MOTR/models/ops/src/vision.cpp:13-16,
    ms_deform_attn.h:21-40).  `im2col_step` is checked like the reference does and otherwise unused (no chunking)."""
    _msda_check((value, spatial_shapes, level_start_index, sampling_loc, attn_weight), value, im2col_step)
    N, S, M, D = value.shape
    _, Lq, _, Lv, P, _ = sampling_loc.shape
    assert spatial_shapes.dtype == torch.int64 and level_start_index.dtype == torch.int64
    fn = {torch.float32: "moy_msda_fwd_f32", torch.bfloat16: "moy_msda_fwd_bf16", torch.float16: "moy_msda_fwd_f16",
          torch.float64: "moy_msda_fwd_f64"}.get(value.dtype)
    if fn is None or sampling_loc.dtype != value.dtype or attn_weight.dtype != value.dtype:
        raise TypeError("ms_deform_attn_forward: float32 / float64 / bfloat16 / float16, one dtype for value, locations and weights")
    out = torch.empty(N, Lq, M * D, device=value.device, dtype=value.dtype)
    L.check(getattr(L.lib(), fn)(value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(), sampling_loc.data_ptr(),
                                 attn_weight.data_ptr(), N, S, M, D, Lv, Lq, P, out.data_ptr(), _st()), fn)
    return out


def ms_deform_attn_backward(value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output, im2col_step=64):
    """MultiScaleDeformableAttention.ms_deform_attn_backward (vision.cpp:13-16, ms_deform_attn.h:42-62):
    -> [grad_value, grad_sampling_loc, grad_attn_weight], fp32 / fp64 like the reference."""
    _msda_check((value, spatial_shapes, level_start_index, sampling_loc, attn_weight, grad_output), value, im2col_step)
    N, S, M, D = value.shape
    _, Lq, _, Lv, P, _ = sampling_loc.shape
    assert spatial_shapes.dtype == torch.int64 and level_start_index.dtype == torch.int64
    fn = {torch.float32: "moy_msda_bwd_f32", torch.float64: "moy_msda_bwd_f64"}.get(value.dtype)
    if fn is None or any(t.dtype != value.dtype for t in (sampling_loc, attn_weight, grad_output)):
        raise TypeError("ms_deform_attn_backward: float32 / float64, one dtype for all floating tensors")
    if grad_output.numel() != N * Lq * M * D:
        raise RuntimeError("grad_output must hold N*Lq*M*D elements")
    gv, gl, ga = torch.empty_like(value), torch.empty_like(sampling_loc), torch.empty_like(attn_weight)
    L.check(getattr(L.lib(), fn)(value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(), sampling_loc.data_ptr(),
                                 attn_weight.data_ptr(), grad_output.data_ptr(), N, S, M, D, Lv, Lq, P, gv.data_ptr(), gl.data_ptr(),
                                 ga.data_ptr(), _st()), fn)
    return [gv, gl, ga]


def resize_linear_u8(frames, out_hw, out=None):
    """uint8 [B, Hs, Ws, 3] -> [B, Hd, Wd, 3], cv2.resize(..., INTER_LINEAR) semantics (LetterBox scaleFill,
    data/augment.py:573-576)."""
    _need_gpu(frames)
    if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[3] != 3 or frames.stride(3) != 1 or frames.stride(2) != 3:
        raise ValueError("frames must be uint8 [B, H, W, 3] with dense pixels")
    B, Hs, Ws, _ = frames.shape
    Hd, Wd = out_hw
    if out is None:
        out = torch.empty(B, Hd, Wd, 3, device=frames.device, dtype=torch.uint8)
    assert out.is_contiguous() and tuple(out.shape) == (B, Hd, Wd, 3) and out.dtype == torch.uint8
    L.check(L.lib().moy_resize_linear_u8(frames.data_ptr(), B, Hs, Ws, frames.stride(1), frames.stride(0), out.data_ptr(), Hd, Wd,
                                         _st()), "moy_resize_linear_u8")
    return out


def box_iou(a, b, na=None, nb=None):
    """a fp32 [T, n, 4], b fp32 [T, K, 4] pixel x0y0x1y1 -> IoU fp32 [T, n, K] (val.py:517-553); na / nb int32 [T]."""
    _need_gpu(a, b)
    T, n, _ = a.shape
    K = b.shape[1]
    out = torch.empty(T, n, K, device=a.device, dtype=torch.float32)
    if n == 0 or K == 0:
        return out
    assert a.dtype == torch.float32 and b.dtype == torch.float32 and a.is_contiguous() and b.is_contiguous()
    L.check(L.lib().moy_box_iou(a.data_ptr(), b.data_ptr(), T, n, K, na.data_ptr() if na is not None else None,
                                nb.data_ptr() if nb is not None else None, out.data_ptr(), _st()), "moy_box_iou")
    return out


def assign_post(logits, boxes, score_thresh=0.4, conf=0.25, img_wh=(1.0, 1.0)):
    _need_gpu(logits, boxes)
    B, nq, nc = logits.shape
    dev = logits.device
    out = dict(y=torch.empty(B, nq, 4 + nc, device=dev), scores=torch.empty(B, nq, device=dev),
               obj_idxes=torch.empty(B, nq, device=dev, dtype=torch.int64), rows=torch.zeros(B, nq, 6, device=dev),
               track_id=torch.full((B, nq), -1, device=dev, dtype=torch.int64),
               n_rows=torch.zeros(B, device=dev, dtype=torch.int32), n_ids=torch.zeros(B, device=dev, dtype=torch.int32))
    L.check(L.lib().moy_assign_post(logits.data_ptr(), boxes.data_ptr(), B, nq, nc, score_thresh, conf, img_wh[0], img_wh[1],
                                    out["y"].data_ptr(), out["scores"].data_ptr(), out["obj_idxes"].data_ptr(),
                                    out["rows"].data_ptr(), out["track_id"].data_ptr(), out["n_rows"].data_ptr(),
                                    out["n_ids"].data_ptr(), _st()), "moy_assign_post")
    return out


def gather_rows(src, rows):
    _need_gpu(src, rows)
    M, N = rows.numel(), src.shape[1]
    out = torch.empty(M, N, device=src.device, dtype=src.dtype)
    L.check(L.lib().moy_gather_rows(src.data_ptr(), _ld(src), rows.data_ptr(), M, N, out.data_ptr(), N, _code(src), _st()),
            "moy_gather_rows")
    return out


def cast_f32_to(src, dtype):
    _need_gpu(src)
    M, N = src.shape
    out = torch.empty(M, N, device=src.device, dtype=dtype)
    L.check(L.lib().moy_cast_f32_to(src.data_ptr(), _ld(src), M, N, out.data_ptr(), N, _code(out), _st()), "moy_cast_f32_to")
    return out


def sigmoid_f32(x):
    _need_gpu(x)
    out = torch.empty_like(x)
    L.check(L.lib().moy_sigmoid_f32(x.data_ptr(), x.numel(), out.data_ptr(), _st()), "moy_sigmoid_f32")
    return out

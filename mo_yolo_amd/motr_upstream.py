"""Upstream-MOTR module surface of SURVEY §8(f) rank 4: the deformable ENCODER stack over the native operator.

Reference classes mirrored (names, constructor arguments, parameter names = `state_dict` keys, `forward` signatures):
  MSDeformAttn                             MOTR/models/ops/modules/ms_deform_attn.py:30-121
  MOTRDeformableTransformerEncoderLayer    MOTR/models/deformable_transformer_plus.py:347-386
  DeformableTransformerEncoder             MOTR/models/deformable_transformer_plus.py:389-415

Every linear is a moy_gemm launch (bias / ReLU / residual + LayerNorm fused), the softmax + sampling-location arithmetic is
moy_msda_prep, the padding mask moy_mask_rows, the sampling itself the operator entry moy_msda_fwd_* (fp32 or bf16) -- the
same C-ABI function `MultiScaleDeformableAttention.ms_deform_attn_forward` binds.  Dropouts are identity (inference).  CPU
tensors raise like the reference's native op ("Not implemented on the CPU").  d_model 256 / any head count dividing it.
"""
from __future__ import annotations

import copy
import ctypes as C

import torch
import torch.nn as nn

from . import _lib as L
from . import ops
from .modules import _LayerNorm, _Linear


class MSDeformAttn(nn.Module):
    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4, sigmoid_attn=False):
        super().__init__()
        if d_model % n_heads != 0:
            raise ValueError("d_model must be divisible by n_heads, but got {} and {}".format(d_model, n_heads))
        if n_levels > 8:
            raise NotImplementedError("at most 8 levels")
        self.im2col_step = 64
        self.sigmoid_attn = sigmoid_attn
        self.d_model, self.n_levels, self.n_heads, self.n_points = d_model, n_levels, n_heads, n_points
        self.sampling_offsets = _Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = _Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = _Linear(d_model, d_model)
        self.output_proj = _Linear(d_model, d_model)

    def sample(self, q2d, pos2d, reference_points, src2d, N, Len_q, Len_in, spatial_shapes, level_start_index, padding_mask):
        """Everything up to (not including) output_proj: rows [N*Len_q, d_model] in the dtype of `src2d`."""
        dev, dt = src2d.device, src2d.dtype
        M, Lv, P = self.n_heads, self.n_levels, self.n_points
        value = self.value_proj.rows(src2d)
        if padding_mask is not None:
            mk = padding_mask.reshape(-1).to(torch.uint8).contiguous()
            L.check(L.lib().moy_mask_rows(value.data_ptr(), value.stride(0), value.shape[0], value.shape[1], mk.data_ptr(),
                                          ops._code(value), ops._st()), "moy_mask_rows")
        no, na = M * Lv * P * 2, M * Lv * P
        offaw = torch.empty(N * Len_q, no + na, device=dev, dtype=torch.float32)
        self.sampling_offsets.rows(q2d, A2=pos2d, out_f32=True, out=offaw[:, :no])
        self.attention_weights.rows(q2d, A2=pos2d, out_f32=True, out=offaw[:, no:])
        ref = reference_points.reshape(N * Len_q, Lv, -1).float().contiguous()
        refdim = ref.shape[-1]
        if refdim not in (2, 4):
            raise ValueError("Last dim of reference_points must be 2 or 4, but get {} instead.".format(refdim))
        sdt = torch.float32 if dt == torch.float32 else torch.bfloat16
        loc = torch.empty(N, Len_q, M, Lv, P, 2, device=dev, dtype=sdt)
        aw = torch.empty(N, Len_q, M, Lv, P, device=dev, dtype=sdt)
        shp = spatial_shapes.detach().cpu().to(torch.int32).contiguous()
        arr = (C.c_int32 * (2 * Lv))(*shp.reshape(-1).tolist())
        L.check(L.lib().moy_msda_prep(offaw.data_ptr(), offaw.stride(0), 0, no, ref.data_ptr(), refdim, N * Len_q, M, Lv, P, arr,
                                      int(bool(self.sigmoid_attn)), loc.data_ptr(), aw.data_ptr(), ops._code(loc), ops._st()),
                "moy_msda_prep")
        v4 = value.view(N, Len_in, M, self.d_model // M)
        if v4.dtype not in (torch.float32, torch.bfloat16):
            v4 = v4.to(torch.bfloat16)
        out = ops.ms_deform_attn_forward(v4, spatial_shapes.to(dev, torch.int64).contiguous(),
                                         level_start_index.to(dev, torch.int64).contiguous(), loc, aw, self.im2col_step)
        return out.view(N * Len_q, self.d_model).to(dt)

    def forward(self, query, reference_points, input_flatten, input_spatial_shapes, input_level_start_index, input_padding_mask=None):
        ops._need_gpu(query, input_flatten)
        N, Len_q, _ = query.shape
        _, Len_in, _ = input_flatten.shape
        assert int((input_spatial_shapes[:, 0] * input_spatial_shapes[:, 1]).sum()) == Len_in
        s = self.sample(query.reshape(N * Len_q, -1).contiguous(), None, reference_points, input_flatten.reshape(N * Len_in, -1).contiguous(),
                        N, Len_q, Len_in, input_spatial_shapes, input_level_start_index, input_padding_mask)
        return self.output_proj.rows(s).view(N, Len_q, self.d_model)


class MOTRDeformableTransformerEncoderLayer(nn.Module):
    def __init__(self, d_model=256, d_ffn=1024, dropout=0.1, activation="relu", n_levels=4, n_heads=8, n_points=4, sigmoid_attn=False):
        super().__init__()
        if activation != "relu" or d_model != 256:
            raise NotImplementedError("relu, d_model 256 (the LayerNorm epilogue of moy_gemm)")
        self.self_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points, sigmoid_attn=sigmoid_attn)
        self.norm1 = _LayerNorm(d_model)
        self.linear1 = _Linear(d_model, d_ffn)
        self.linear2 = _Linear(d_ffn, d_model)
        self.norm2 = _LayerNorm(d_model)

    @staticmethod
    def with_pos_embed(tensor, pos):
        return tensor if pos is None else tensor + pos

    def rows_forward(self, src2d, pos2d, reference_points, N, S, spatial_shapes, level_start_index, padding_mask):
        a = self.self_attn
        s = a.sample(src2d, pos2d, reference_points, src2d, N, S, S, spatial_shapes, level_start_index, padding_mask)
        src2d = a.output_proj.rows(s, R=src2d, ln=self.norm1.pair(src2d.device))          # norm1(src + self_attn(...))
        h = self.linear1.rows(src2d, act=L.ACT_RELU)
        return self.linear2.rows(h, R=src2d, ln=self.norm2.pair(src2d.device))            # norm2(src + ffn(src))

    def forward(self, src, pos, reference_points, spatial_shapes, level_start_index, padding_mask=None):
        ops._need_gpu(src)
        N, S, Cc = src.shape
        p2 = None if pos is None else pos.reshape(N * S, Cc).to(src.dtype).contiguous()
        return self.rows_forward(src.reshape(N * S, Cc).contiguous(), p2, reference_points, N, S, spatial_shapes, level_start_index,
                                 padding_mask).view(N, S, Cc)


class DeformableTransformerEncoder(nn.Module):
    def __init__(self, encoder_layer, num_layers):
        super().__init__()
        self.layers = nn.ModuleList([copy.deepcopy(encoder_layer) for _ in range(num_layers)])
        self.num_layers = num_layers

    @staticmethod
    def get_reference_points(spatial_shapes, valid_ratios, device):
        """deformable_transformer_plus.py:395-408 (input independent up to valid_ratios; tiny host-side set-up)."""
        reference_points_list = []
        for lvl, (H_, W_) in enumerate(spatial_shapes.tolist()):
            ref_y, ref_x = torch.meshgrid(torch.linspace(0.5, H_ - 0.5, H_, dtype=torch.float32, device=device),
                                          torch.linspace(0.5, W_ - 0.5, W_, dtype=torch.float32, device=device), indexing="ij")
            ref_y = ref_y.reshape(-1)[None] / (valid_ratios[:, None, lvl, 1] * H_)
            ref_x = ref_x.reshape(-1)[None] / (valid_ratios[:, None, lvl, 0] * W_)
            reference_points_list.append(torch.stack((ref_x, ref_y), -1))
        reference_points = torch.cat(reference_points_list, 1)
        return reference_points[:, :, None] * valid_ratios[:, None]

    def forward(self, src, spatial_shapes, level_start_index, valid_ratios, pos=None, padding_mask=None):
        ops._need_gpu(src)
        N, S, Cc = src.shape
        reference_points = self.get_reference_points(spatial_shapes, valid_ratios.to(src.device).float(), device=src.device)
        out = src.reshape(N * S, Cc).contiguous()
        p2 = None if pos is None else pos.reshape(N * S, Cc).to(src.dtype).contiguous()
        for layer in self.layers:
            out = layer.rows_forward(out, p2, reference_points, N, S, spatial_shapes, level_start_index, padding_mask)
        return out.view(N, S, Cc)

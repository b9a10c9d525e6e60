"""Drop-in module surface: the reference's class names, constructor arguments, parameter names
(`state_dict` keys) and `forward()` signatures for the hot path, computed by libmoyolo.so.

Reference classes mirrored (SURVEY §8b "Module forward() surface"):
  Conv / Concat            ultralytics/nn/modules/conv.py:25-42, 287-297
  Bottleneck / C2f / SPPF  ultralytics/nn/modules/block.py:271-283, 168-188, 119-134
  MLP / MSDeformAttn / MOTRDecoderLayer / MOTRTransformerDecoder
                           ultralytics/nn/modules/transformer.py:149-161, 193-287, 515-652, 663-728
  MYDecoder / MOTRTrack    ultralytics/nn/modules/head.py:807-1137, 90-513
  QueryInteractionModule   MOTR/models/qim.py:73-340      Instances  MOTR/models/structures/instances.py
  TrackingModel.predict    ultralytics/nn/tasks.py:486-514
  multi_scale_deformable_attn_pytorch   ultralytics/nn/modules/utils.py:41-78

Logical tensor shapes are the reference's (NCHW feature maps, [bs, L, C] tokens); storage is
channels-last so the kernels see [pixels, C] rows.  The nn.Module containers exist to hold the
parameters under the reference's names; the math never touches torch ops on the compute path.
Fine-grained modules (Conv ... MOTRDecoderLayer, QIM) call the C ABI per op; MYDecoder, MOTRTrack and
TrackingModel run pre-planned TrackEngine launch lists.  CPU tensors raise (no fallback).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import _lib as L
from . import ops
from .config import TrackArch, build_arch

BN_EPS = 1e-3        # utils/torch_utils.py:262 (initialize_weights)


# ----------------------------------------------------------------------------- layout helpers
def _rows(x: torch.Tensor):
    """NCHW-shaped tensor -> ([B*H*W, C] channels-last rows, B, H, W)."""
    ops._need_gpu(x)
    B, Cc, H, W = x.shape
    return x.permute(0, 2, 3, 1).contiguous().view(B * H * W, Cc), B, H, W


def _nchw(y2d: torch.Tensor, B, H, W):
    return y2d.view(B, H, W, -1).permute(0, 3, 1, 2)


class _Prepared:
    """Device-side re-layout of a module's parameters, rebuilt when dtype/device change or after
    load_state_dict (cleared by the hook registered in _ParamModule)."""

    def __init__(self):
        self.key = None
        self.data = None


class _ParamModule(nn.Module):
    def __init__(self):
        super().__init__()
        object.__setattr__(self, "_prep", _Prepared())
        self.register_load_state_dict_post_hook(lambda m, _k: m._invalidate())

    def _invalidate(self):
        self._prep.key = None

    def _prepared(self, dev, dtype, build):
        key = (str(dev), dtype)
        if self._prep.key != key:
            with torch.no_grad():
                self._prep.data = build(dev, dtype)
            self._prep.key = key
        return self._prep.data


def _bn_fold(bn: nn.BatchNorm2d, dev):
    scale = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).float()
    shift = (bn.bias - bn.running_mean * scale).float()
    return scale.to(dev).contiguous(), shift.to(dev).contiguous()


# ----------------------------------------------------------------------------- conv blocks
class Conv(_ParamModule):
    """conv.py:25-42.  forward(x) = SiLU(BN(conv2d(x))) in one fused launch."""

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, d=1, act=True):
        super().__init__()
        if g != 1 or d != 1 or k not in (1, 3) or (p is not None and p != k // 2):
            raise NotImplementedError("hot path uses k in {1,3}, groups 1, dilation 1, autopad")
        self.conv = nn.Conv2d(c1, c2, k, s, k // 2, bias=False)
        self.bn = nn.BatchNorm2d(c2, eps=BN_EPS, momentum=0.03)
        self.act_code = L.ACT_SILU if act is True else L.ACT_NONE
        self.c1, self.c2, self.k, self.s = c1, c2, k, s

    def _build(self, dev, dtype):
        w = self.conv.weight.detach().float()
        scale, shift = _bn_fold(self.bn, dev)
        if self.c1 == 3:      # stem: fused with the float-tensor preprocess branch
            return dict(stem=w.permute(2, 3, 1, 0).reshape(27, self.c2).contiguous().to(dev), scale=scale, shift=shift)
        w2 = w.reshape(self.c2, self.c1) if self.k == 1 else w.permute(0, 2, 3, 1).reshape(self.c2, 9 * self.c1)
        return dict(w=ops.pad_weight(w2.to(dev), dtype), scale=scale, shift=shift)

    def rows_forward(self, x2d, B, H, W, out=None, residual=None, dtype=None):
        """Channels-last rows in, rows out (used by C2f/SPPF to write concat slices in place)."""
        dtype = dtype or x2d.dtype
        pr = self._prepared(x2d.device, dtype, self._build)
        if self.k == 1:
            y = ops.gemm(x2d, pr["w"], self.c2, self.c1, out=out, scale=pr["scale"], shift=pr["shift"], act=self.act_code,
                         R=residual)
            return y, H, W
        Ho, Wo = (H + 2 - 3) // self.s + 1, (W + 2 - 3) // self.s + 1
        y = ops.gemm(x2d, pr["w"], self.c2, 9 * self.c1, out=out, ksize=3, stride=self.s, geom=(B, H, W, Ho, Wo, self.c1),
                     scale=pr["scale"], shift=pr["shift"], act=self.act_code, R=residual)
        return y, Ho, Wo

    def forward(self, x):
        ops._need_gpu(x)
        if self.c1 == 3:
            assert self.k == 3 and self.s == 2
            B, _, H, W = x.shape
            dtype = x.dtype if x.dtype == torch.bfloat16 else torch.float32
            pr = self._prepared(x.device, dtype, self._build)
            y = ops.stem_conv(x.float().contiguous(), pr["stem"], pr["scale"], pr["shift"], dtype)
            return _nchw(y, B, H // 2, W // 2)
        x2d, B, H, W = _rows(x)
        y, Ho, Wo = self.rows_forward(x2d, B, H, W)
        return _nchw(y, B, Ho, Wo)

    forward_fuse = forward      # conv.py:40: BN is always folded here


class Concat(nn.Module):
    """conv.py:287-297."""

    def __init__(self, dimension=1):
        super().__init__()
        self.d = dimension

    def forward(self, x):
        return torch.cat(x, self.d)


class Bottleneck(nn.Module):
    """block.py:271-283."""

    def __init__(self, c1, c2, shortcut=True, g=1, k=(3, 3), e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, k[0] if isinstance(k[0], int) else k[0][0], 1)
        self.cv2 = Conv(c_, c2, k[1] if isinstance(k[1], int) else k[1][0], 1, g=g)
        self.add = shortcut and c1 == c2

    def rows_forward(self, x2d, B, H, W, out=None):
        t, _, _ = self.cv1.rows_forward(x2d, B, H, W)
        y, _, _ = self.cv2.rows_forward(t, B, H, W, out=out, residual=x2d if self.add else None)
        return y

    def forward(self, x):
        x2d, B, H, W = _rows(x)
        return _nchw(self.rows_forward(x2d, B, H, W), B, H, W)


class C2f(nn.Module):
    """block.py:168-188: chunk / bottleneck chain / cat are channel slices of one buffer."""

    def __init__(self, c1, c2, n=1, shortcut=False, g=1, e=0.5):
        super().__init__()
        self.c = int(c2 * e)
        self.cv1 = Conv(c1, 2 * self.c, 1, 1)
        self.cv2 = Conv((2 + n) * self.c, c2, 1)
        self.m = nn.ModuleList(Bottleneck(self.c, self.c, shortcut, g, k=((3, 3), (3, 3)), e=1.0) for _ in range(n))

    def forward(self, x):
        x2d, B, H, W = _rows(x)
        c, n = self.c, len(self.m)
        cat = torch.empty(x2d.shape[0], (2 + n) * c, device=x2d.device, dtype=x2d.dtype)
        self.cv1.rows_forward(x2d, B, H, W, out=cat[:, :2 * c])
        for j, m in enumerate(self.m):
            m.rows_forward(cat[:, (1 + j) * c:(2 + j) * c], B, H, W, out=cat[:, (2 + j) * c:(3 + j) * c])
        y, _, _ = self.cv2.rows_forward(cat, B, H, W)
        return _nchw(y, B, H, W)


class SPPF(nn.Module):
    """block.py:119-134 (k = 5 only: the cascaded pools are windows 5/9/13)."""

    def __init__(self, c1, c2, k=5):
        super().__init__()
        if k != 5:
            raise NotImplementedError("SPPF k=5")
        c_ = c1 // 2
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_ * 4, c2, 1, 1)

    def forward(self, x):
        x2d, B, H, W = _rows(x)
        c_ = self.cv1.c2
        cat = torch.empty(x2d.shape[0], 4 * c_, device=x2d.device, dtype=x2d.dtype)
        self.cv1.rows_forward(x2d, B, H, W, out=cat[:, :c_])
        lib = L.lib()
        esz = cat.element_size()
        L.check(lib.moy_sppf_pool(cat.data_ptr(), 4 * c_, B, H, W, c_, cat.data_ptr() + c_ * esz, cat.data_ptr() + 2 * c_ * esz,
                                  cat.data_ptr() + 3 * c_ * esz, 4 * c_, ops._code(cat), ops._st()), "moy_sppf_pool")
        y, _, _ = self.cv2.rows_forward(cat, B, H, W)
        return _nchw(y, B, H, W)


# ----------------------------------------------------------------------------- transformer pieces
class _Linear(_ParamModule):
    """nn.Linear parameter holder (`weight`, `bias`) evaluated by moy_gemm / moy_rowdot."""

    def __init__(self, cin, cout):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin))
        self.bias = nn.Parameter(torch.zeros(cout))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        self.cin, self.cout = cin, cout

    def _build(self, dev, dtype):
        return dict(w=ops.pad_weight(self.weight.detach().to(dev), dtype), b=self.bias.detach().float().to(dev).contiguous(),
                    wf=self.weight.detach().float().to(dev).contiguous())

    def rows(self, x2d, act=L.ACT_NONE, A2=None, R=None, ln=None, out_f32=False, out=None, a_rows=None):
        pr = self._prepared(x2d.device, x2d.dtype, self._build)
        return ops.gemm(x2d, pr["w"], self.cout, self.cin, shift=pr["b"], act=act, A2=A2, R=R, ln=ln, out_f32=out_f32, out=out,
                        a_rows=a_rows)

    def narrow(self, x2d, mode=0, aux=None, aux_rows=None):
        pr = self._prepared(x2d.device, x2d.dtype, self._build)
        return ops.rowdot(x2d, pr["wf"], pr["b"], mode=mode, aux=aux, aux_rows=aux_rows)

    def forward(self, x):
        shp = x.shape
        y = self.rows(x.reshape(-1, shp[-1]).contiguous()) if self.cout % 4 == 0 and self.cout > 8 else \
            self.narrow(x.reshape(-1, shp[-1]).contiguous())
        return y.view(*shp[:-1], self.cout)


class _LayerNorm(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))

    def pair(self, dev):
        return (self.weight.detach().float().to(dev).contiguous(), self.bias.detach().float().to(dev).contiguous())


class MLP(nn.Module):
    """transformer.py:149-161 (3-layer 256->256->256->4 box heads on this path)."""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(_Linear(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))

    def rows(self, x2d, mode=0, aux=None, aux_rows=None, a_rows=None):
        for i, layer in enumerate(self.layers[:-1]):
            x2d = layer.rows(x2d, act=L.ACT_RELU, a_rows=a_rows if i == 0 else None)
        return self.layers[-1].narrow(x2d, mode=mode, aux=aux, aux_rows=aux_rows)

    def forward(self, x):
        shp = x.shape
        return self.rows(x.reshape(-1, shp[-1]).contiguous()).view(*shp[:-1], -1)


def pos2posemb(pos, num_pos_feats=64, temperature=10000):
    """transformer.py:183-190."""
    if num_pos_feats != 64 or temperature != 10000:
        raise NotImplementedError
    shp = pos.shape
    return ops.pos2posemb(pos.reshape(-1, 4).float().contiguous(), torch.float32).view(*shp[:-1], 256)


def inverse_sigmoid(x, eps=1e-5):
    """nn/modules/utils.py:34-38 (tiny elementwise helper kept in torch: not on the planned path)."""
    x = x.clamp(min=0, max=1)
    return torch.log(x.clamp(min=eps) / (1 - x).clamp(min=eps))


def multi_scale_deformable_attn_pytorch(value, value_spatial_shapes, sampling_locations, attention_weights):
    """Same name/arguments as nn/modules/utils.py:41; dispatches to the native operator."""
    shapes = torch.as_tensor([list(s) for s in value_spatial_shapes], dtype=torch.int64, device=value.device)
    lsi = torch.cat((shapes.new_zeros(1), (shapes[:, 0] * shapes[:, 1]).cumsum(0)[:-1]))
    return ops.ms_deform_attn_forward(value.contiguous(), shapes, lsi, sampling_locations.contiguous(),
                                      attention_weights.contiguous(), 64)


class MSDeformAttnFunction(torch.autograd.Function):
    """MOTR/models/ops/functions/ms_deform_attn_func.py:24-41: the autograd wrapper of the native op
    (forward + `ms_deform_attn_backward`); gradients for value, sampling locations and attention weights."""

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights, im2col_step):
        ctx.im2col_step = im2col_step
        output = ops.ms_deform_attn_forward(value, value_spatial_shapes, value_level_start_index, sampling_locations,
                                            attention_weights, ctx.im2col_step)
        ctx.save_for_backward(value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights)
        return output

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_output):
        value, shapes, lsi, loc, aw = ctx.saved_tensors
        gv, gl, ga = ops.ms_deform_attn_backward(value, shapes, lsi, loc, aw, grad_output.contiguous(), ctx.im2col_step)
        return gv, None, None, gl, ga, None


class MSDeformAttn(nn.Module):
    """transformer.py:193-287 (d_model 256, 8 heads, 4 points, <= 4 levels, 4-d reference boxes)."""

    def __init__(self, d_model=256, n_levels=4, n_heads=8, n_points=4, my_softmax=False):
        super().__init__()
        if d_model != 256 or n_heads != 8 or n_points != 4 or n_levels > 4:
            raise NotImplementedError("kernels are specialised to d_model 256 / 8 heads / 4 points / <=4 levels")
        self.im2col_step, self.d_model, self.n_levels, self.n_heads, self.n_points = 64, d_model, n_levels, n_heads, n_points
        self.sampling_offsets = _Linear(d_model, n_heads * n_levels * n_points * 2)
        self.attention_weights = _Linear(d_model, n_heads * n_levels * n_points)
        self.value_proj = _Linear(d_model, d_model)
        self.output_proj = _Linear(d_model, d_model)

    def core(self, q2d, ref2d, value2d, bs, len_q, shapes):
        """offsets+weights projection -> fused sampling -> rows [bs*len_q, 256] (before output_proj)."""
        offaw = torch.empty(bs * len_q, self.n_heads * self.n_levels * self.n_points * 3, device=q2d.device)
        no = self.sampling_offsets.cout
        self.sampling_offsets.rows(q2d, out_f32=True, out=offaw[:, :no])
        self.attention_weights.rows(q2d, out_f32=True, out=offaw[:, no:])
        S = value2d.shape[0] // bs
        return ops.msda_fused(value2d, bs, S, shapes, offaw, ref2d, len_q)

    def forward(self, query, refer_bbox, value, value_shapes, value_mask=None):
        bs, len_q = query.shape[:2]
        if refer_bbox.shape[-1] != 4:
            raise NotImplementedError("4-d reference boxes (transformer.py:280-282)")
        v2d = self.value_proj.rows(value.reshape(-1, self.d_model).contiguous())
        if value_mask is not None:
            # transformer.py:258-259: `value = value.masked_fill(value_mask[..., None], float(0))` on the PROJECTED value (always None
            # on the tracking path; kept for callers of the module surface)
            mk = value_mask.reshape(-1).to(device=v2d.device, dtype=torch.uint8).contiguous()
            if mk.numel() != v2d.shape[0]:
                raise ValueError(f"value_mask covers {mk.numel()} tokens, value has {v2d.shape[0]}")
            L.check(L.lib().moy_mask_rows(v2d.data_ptr(), v2d.stride(0), v2d.shape[0], v2d.shape[1], mk.data_ptr(), ops._code(v2d), ops._st()),
                    "moy_mask_rows")
        ref2d = refer_bbox.reshape(bs * len_q, -1, 4)[:, 0].float().contiguous()   # same box for every level (:644)
        samp = self.core(query.reshape(-1, self.d_model).contiguous(), ref2d, v2d, bs, len_q, [tuple(s) for s in value_shapes])
        return self.output_proj.rows(samp).view(bs, len_q, self.d_model)


class _MHA(_ParamModule):
    """nn.MultiheadAttention parameter names: in_proj_weight/in_proj_bias/out_proj.{weight,bias}."""

    def __init__(self, E, nh):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * E, E))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * E))
        self.out_proj = _Linear(E, E)
        nn.init.xavier_uniform_(self.in_proj_weight)
        self.E, self.nh = E, nh

    def _build(self, dev, dtype):
        W, b, E = self.in_proj_weight.detach(), self.in_proj_bias.detach().float(), self.E
        return dict(wqk=ops.pad_weight(W[:2 * E].to(dev), dtype), bqk=b[:2 * E].to(dev).contiguous(),
                    wv=ops.pad_weight(W[2 * E:].to(dev), dtype), bv=b[2 * E:].to(dev).contiguous())

    def attend(self, x2d, pos2d, bs, Lq, R=None, ln=None):
        """q = k = x + pos, v = x (transformer.py:637-639); returns out_proj(attn) [+R, LN]."""
        pr = self._prepared(x2d.device, x2d.dtype, self._build)
        E = self.E
        qkv = torch.empty(bs * Lq, 3 * E, device=x2d.device, dtype=x2d.dtype)
        ops.gemm(x2d, pr["wqk"], 2 * E, E, shift=pr["bqk"], A2=pos2d, out=qkv[:, :2 * E])
        ops.gemm(x2d, pr["wv"], E, E, shift=pr["bv"], out=qkv[:, 2 * E:])
        a = ops.mha_core(qkv, bs, Lq, self.nh)
        return self.out_proj.rows(a, R=R, ln=ln)


class MOTRDecoderLayer(nn.Module):
    """transformer.py:515-652 (eval: dropouts are identity)."""

    def __init__(self, d_model=256, n_heads=8, d_ffn=1024, dropout=0.1, act=None, n_levels=4, n_points=4, **_unused):
        super().__init__()
        self.self_attn = _MHA(d_model, n_heads)
        self.norm1 = _LayerNorm(d_model)
        self.cross_attn = MSDeformAttn(d_model, n_levels, n_heads, n_points)
        self.norm2 = _LayerNorm(d_model)
        self.linear1 = _Linear(d_model, d_ffn)
        self.linear2 = _Linear(d_ffn, d_model)
        self.norm3 = _LayerNorm(d_model)
        self.d_model = d_model

    def rows_forward(self, e2d, ref2d, value2d, shapes, pos2d, bs, Lq):
        dev = e2d.device
        e1 = self.self_attn.attend(e2d, pos2d, bs, Lq, R=e2d, ln=self.norm1.pair(dev))
        # cross attention on (e1 + pos): the offsets/weights GEMMs take the prologue add
        ca = self.cross_attn
        offaw = torch.empty(bs * Lq, ca.n_heads * ca.n_levels * ca.n_points * 3, device=dev)
        no = ca.sampling_offsets.cout
        ca.sampling_offsets.rows(e1, A2=pos2d, out_f32=True, out=offaw[:, :no])
        ca.attention_weights.rows(e1, A2=pos2d, out_f32=True, out=offaw[:, no:])
        samp = ops.msda_fused(value2d, bs, value2d.shape[0] // bs, shapes, offaw, ref2d, Lq)
        e2 = ca.output_proj.rows(samp, R=e1, ln=self.norm2.pair(dev))
        h = self.linear1.rows(e2, act=L.ACT_RELU)
        return self.linear2.rows(h, R=e2, ln=self.norm3.pair(dev))

    def forward(self, embed, refer_bbox, feats, shapes, padding_mask=None, attn_mask=None, track_query_pos=None):
        if padding_mask is not None or attn_mask is not None:
            raise NotImplementedError("masks are None on the inference path")
        bs, Lq, Cc = embed.shape
        v2d = self.cross_attn.value_proj.rows(feats.reshape(-1, Cc).contiguous())
        out = self.rows_forward(embed.reshape(-1, Cc).contiguous(), refer_bbox.reshape(-1, 4).float().contiguous(), v2d,
                                [tuple(s) for s in shapes], track_query_pos.reshape(-1, Cc).to(embed.dtype).contiguous(), bs, Lq)
        return out.view(bs, Lq, Cc)


class MOTRTransformerDecoder(nn.Module):
    """transformer.py:663-728, eval branch: iterative refinement, scores from the last layer only."""

    def __init__(self, hidden_dim, decoder_layer, num_layers, eval_idx=-1):
        super().__init__()
        import copy
        self.layers = nn.ModuleList(copy.deepcopy(decoder_layer) for _ in range(num_layers))
        self.num_layers, self.hidden_dim = num_layers, hidden_dim
        self.eval_idx = eval_idx if eval_idx >= 0 else num_layers + eval_idx

    def forward(self, embed, refer_bbox, feats, shapes, bbox_head, score_head, pos_mlp, attn_mask=None, padding_mask=None,
                track_query_embed=None):
        bs, Lq, Cc = embed.shape
        shp = [tuple(s) for s in shapes]
        e = embed.reshape(-1, Cc).contiguous()
        pos = track_query_embed.reshape(-1, Cc).to(embed.dtype).contiguous()
        f2d = feats.reshape(-1, Cc).contiguous()
        ref = ops.sigmoid_f32(refer_bbox.reshape(-1, 4).float().contiguous())
        for i, layer in enumerate(self.layers):
            v2d = layer.cross_attn.value_proj.rows(f2d)
            e = layer.rows_forward(e, ref, v2d, shp, pos, bs, Lq)
            ref = bbox_head[i].rows(e, mode=1, aux=ref)         # sigmoid(bbox_head(out) + inverse_sigmoid(ref)), :709
            if i == self.eval_idx:
                scores = score_head[i].narrow(e) if score_head[i].cout <= 8 else score_head[i].rows(e, out_f32=True)
                return ref.view(1, bs, Lq, 4), scores.view(1, bs, Lq, -1), e.view(bs, Lq, Cc)
        raise RuntimeError("eval_idx beyond the last layer")


# ----------------------------------------------------------------------------- Instances / QIM
class Instances:
    """Field container with the subset of MOTR/models/structures/instances.py the path uses.
    Boolean-mask indexing is a device gather (the reference loops per element, :135-190)."""

    def __init__(self, image_size=(1, 1), **fields):
        object.__setattr__(self, "_image_size", image_size)
        object.__setattr__(self, "_fields", {})
        for k, v in fields.items():
            self.set(k, v)

    def __setattr__(self, name, val):
        if name.startswith("_"):
            object.__setattr__(self, name, val)
        else:
            self.set(name, val)

    def __getattr__(self, name):
        f = object.__getattribute__(self, "_fields")
        if name not in f:
            raise AttributeError(f"Cannot find field '{name}' in the given Instances!")
        return f[name]

    def set(self, name, value):
        self._fields[name] = value

    def has(self, name):
        return name in self._fields

    def get(self, name):
        return self._fields[name]

    def get_fields(self):
        return self._fields

    def __len__(self):
        for v in self._fields.values():
            return len(v)
        raise NotImplementedError("Empty Instances does not support __len__!")

    def to(self, *a, **k):
        return Instances(self._image_size, **{n: (v.to(*a, **k) if hasattr(v, "to") else v) for n, v in self._fields.items()})

    def __getitem__(self, item):
        if isinstance(item, torch.Tensor) and item.dtype == torch.bool:
            item = item.reshape(-1)
        return Instances(self._image_size, **{n: v[item] for n, v in self._fields.items()})

    @staticmethod
    def cat(lst: List["Instances"]):
        return Instances(lst[0]._image_size, **{k: torch.cat([i.get(k) for i in lst], 0) for k in lst[0]._fields})


class QueryInteractionModule(nn.Module):
    """MOTR/models/qim.py:73-340.  `forward` keeps the shipped behaviour (returns track_queries
    unchanged, qim.py:314-340); `_update_track_embedding` (qim.py:251-301) is the learned update."""

    def __init__(self, args=None, dim_in=256, hidden_dim=256, dim_out=512):
        super().__init__()
        self.self_attn = _MHA(dim_in, 8)
        self.linear1 = _Linear(dim_in, hidden_dim)
        self.linear2 = _Linear(hidden_dim, dim_in)
        self.linear_feat1 = _Linear(dim_in, hidden_dim)
        self.linear_feat2 = _Linear(hidden_dim, dim_in)
        self.norm_feat = _LayerNorm(dim_in)
        self.norm1 = _LayerNorm(dim_in)
        self.norm2 = _LayerNorm(dim_in)

    def _update_track_embedding(self, track_instances: Instances) -> Instances:
        n = len(track_instances)
        if n == 0:
            return track_instances
        out_embed = track_instances.output_embedding.contiguous()
        dev, dt = out_embed.device, out_embed.dtype
        qpos = ops.pos2posemb(track_instances.ref_pts.float().contiguous(), dt)
        tgt = self.self_attn.attend(out_embed, qpos, 1, n, R=out_embed, ln=self.norm1.pair(dev))
        h = self.linear1.rows(tgt, act=L.ACT_RELU)
        tgt = self.linear2.rows(h, R=tgt, ln=self.norm2.pair(dev))
        h = self.linear_feat1.rows(tgt, act=L.ACT_RELU)
        qf = self.linear_feat2.rows(h, R=track_instances.query_pos.to(dt).contiguous(), ln=self.norm_feat.pair(dev))
        track_instances.query_pos = qf
        track_instances.ref_pts = inverse_sigmoid(track_instances.pred_boxes[:, :4].detach().clone())
        return track_instances

    def forward(self, data: dict) -> Instances:
        return data["track_queries"]


# ----------------------------------------------------------------------------- planned modules
class MYDecoder(nn.Module):
    """head.py:807-1137.  Parameters under the reference names; `forward` (head.py:873-985) runs the head part of a
    TrackEngine plan built for the incoming feature-map shapes and returns the reference's 7-tuple."""

    def __init__(self, nc=80, ch=(512, 1024, 2048), hd=256, nq=300, ndp=4, nh=8, ndl=6, d_ffn=1024, **_unused):
        super().__init__()
        self.hidden_dim, self.nhead, self.nl, self.nc, self.num_queries, self.num_decoder_layers = hd, nh, len(ch), nc, nq, ndl
        self.ch = tuple(ch)
        self.input_proj = nn.ModuleList(nn.Sequential(nn.Conv2d(x, hd, 1, bias=False), nn.BatchNorm2d(hd, eps=BN_EPS)) for x in ch)
        layer = MOTRDecoderLayer(hd, nh, d_ffn, 0.0, None, self.nl, ndp)
        self.decoder = MOTRTransformerDecoder(hd, layer, ndl, -1)
        self.denoising_class_embed = nn.Embedding(nc, hd)
        self.query_pos_head = MLP(4, 2 * hd, hd, num_layers=2)
        self.enc_output = nn.Sequential(_Linear(hd, hd), _LayerNorm(hd))
        self.enc_score_head = _Linear(hd, nc)
        self.enc_bbox_head = MLP(hd, hd, 4, num_layers=3)
        self.dec_score_head = nn.ModuleList([_Linear(hd, nc) for _ in range(ndl)])
        self.dec_bbox_head = nn.ModuleList([MLP(hd, hd, 4, num_layers=3) for _ in range(ndl)])
        self._engines: Dict = {}
        self.register_load_state_dict_post_hook(lambda m, _k: m._engines.clear())

    def _engine(self, feats):
        """Head-only TrackEngine for these pyramid shapes (cached; cleared when weights are loaded)."""
        from .engine import TrackEngine
        B = feats[0].shape[0]
        shapes = tuple((f.shape[2], f.shape[3]) for f in feats)
        dt = feats[0].dtype if feats[0].dtype in (torch.bfloat16, torch.float16) else torch.float32
        key = (B, shapes, dt, str(feats[0].device))
        if key not in self._engines:
            sd = {f"model.0.decoder.{k}": v for k, v in self.state_dict().items()}
            arch = TrackArch(nc=self.nc, nq=self.num_queries, layers=[], head_ch=tuple(self.ch))
            self._engines[key] = TrackEngine(arch, sd, shapes[0][0] * 8, shapes[0][1] * 8, batch=B, dtype=dt,
                                             device=feats[0].device, head_only=True, level_shapes_override=shapes,
                                             scale_boxes=False)
        return self._engines[key]

    def forward(self, x, track_ref_pts=None, batch=None, is_first=False, pre_class=None, track_query_pos=None):
        """x = [P3, P4, P5] NCHW device tensors -> (dec_bboxes [1,bs,nq,4], dec_scores [1,bs,nq,nc], enc_bboxes [bs,nq,4],
        enc_scores [bs,nq,nc], dn_meta = None, init_reference [bs,nq,4], dec_output_embedding [bs,nq,256]) as head.py:985.
        The carried-track arguments select a branch that cannot run in the reference (IndexError at head.py:1235);
        the carried design lives in TrackEngine(temporal=...) / TrackingModel.inference_single_image."""
        if track_ref_pts is not None or pre_class is not None or track_query_pos is not None:
            raise NotImplementedError("carried track queries: use TrackEngine(temporal=n) / TrackingModel.inference_single_image "
                                      "(the reference's own branch for these arguments raises, head.py:1235)")
        ops._need_gpu(*x)
        eng = self._engine(x)
        out = eng.forward_head(x)
        return _x7(eng, out)


def _x7(eng, out):
    """The reference's 7-tuple (head.py:985) from engine outputs; fresh tensors (engine buffers are static)."""
    B = eng.B
    refer = out["refer_bbox_logit"]
    enc_scores = eng.scores_all.view(B, eng.S, -1)[torch.arange(B, device=refer.device)[:, None], out["topk_ind"].long()]
    enc_boxes = ops.sigmoid_f32(refer.contiguous())                   # enc_bboxes = sigmoid(refer_bbox) (head.py:1100), on the library
    return (out["boxes"].unsqueeze(0).clone(), out["logits"].unsqueeze(0).clone(), enc_boxes, enc_scores, None,
            enc_boxes.clone(), out["hs"].clone())


class MOTRTrack(nn.Module):
    """head.py:90-513.  forward(x=[P3,P4,P5]) -> ((y, x7), Instances) with the shipped per-frame
    reset semantics; owns mutable `track_instances` (one instance per sequence, not thread-safe)."""

    def __init__(self, nc=80, ch=(), d_model=256, aux_loss=False, nq=300):
        super().__init__()
        self.nc, self.nl, self.nq = nc, len(ch), nq
        self.decoder = MYDecoder(nc=nc, ch=ch, nq=nq)
        self.track_embed = QueryInteractionModule(None, d_model, self.decoder.hidden_dim, d_model * 2)
        self.track_instances = None

    def forward(self, x, batch=None, is_first=True):
        ops._need_gpu(*x)
        eng = self.decoder._engine(x)            # one plan serves MYDecoder.forward and this (cleared when weights are loaded)
        out = eng.forward_head(x)
        return _reference_outputs(eng, out, self)


def _reference_outputs(eng, out, owner):
    """Pack engine outputs into the reference's eval return structure (head.py:235-239):
    ((y, x7), Instances) with x7 = (dec_bboxes, dec_scores, enc_bboxes, enc_scores, dn_meta,
    init_reference, dec_output_embedding).  Everything returned is a FRESH tensor per call, like the reference's (only raw
    `TrackEngine.outputs()` aliases the engine's static buffers, which the next step or graph replay overwrites).  The
    Instances carry the 11 fields of `_generate_empty_tracks` (head.py:150-189); the five the shipped path never writes keep
    their initial values (zeros / -1), `ref_pts` holds the queries' reference boxes in logit space (the reference leaves
    `torch.rand` there, head.py:163: unused, not reproducible)."""
    B, nq = eng.B, eng.arch.nq
    y = out["y"].clone()
    x7 = _x7(eng, out)
    refer = out["refer_bbox_logit"]
    dev = y.device
    insts = []
    for b in range(B):
        inst = Instances((1, 1))
        inst.ref_pts = refer[b].clone()
        inst.query_pos = torch.zeros(nq, 256, dtype=torch.float32, device=dev)
        inst.output_embedding = out["hs"][b].clone()
        inst.obj_idxes = out["obj_idxes"][b].unsqueeze(1).clone()
        inst.matched_gt_idxes = torch.full((nq,), -1, dtype=torch.long, device=dev)
        inst.disappear_time = torch.zeros(nq, 1, dtype=torch.long, device=dev)
        inst.iou = torch.zeros(nq, dtype=torch.float32, device=dev)
        inst.scores = out["scores"][b].clone()
        inst.track_scores = torch.zeros(nq, 4, dtype=torch.float32, device=dev)
        inst.pred_boxes = out["boxes"][b].clone()
        inst.pred_logits = out["logits"][b].clone()
        insts.append(inst)
    owner.track_instances = insts[0] if B == 1 else insts     # the reference is batch-1 (head.py:235)
    return (y, x7), owner.track_instances


class TrackingModel(nn.Module):
    """ultralytics/nn/tasks.py:299-514, inference surface: `predict(x)` / `forward(x)`.
    The layer list mirrors parse_model on yolo_track.yaml so `state_dict()` keys match the reference."""

    def __init__(self, depth=0.33, width=0.50, nc=1, nq=300):
        super().__init__()
        self.arch = build_arch(depth, width, nc, nq)
        layers: List[nn.Module] = []
        for Ls in self.arch.layers:
            if Ls.kind == "Conv":
                layers.append(Conv(Ls.c1, Ls.c2, Ls.k, Ls.s))
            elif Ls.kind == "C2f":
                layers.append(C2f(Ls.c1, Ls.c2, Ls.n, Ls.shortcut))
            elif Ls.kind == "SPPF":
                layers.append(SPPF(Ls.c1, Ls.c2, Ls.k))
            elif Ls.kind == "Upsample":
                layers.append(nn.Upsample(None, 2, "nearest"))
            else:
                layers.append(Concat(1))
        layers.append(MOTRTrack(nc, self.arch.head_ch, nq=nq))
        self.model = nn.Sequential(*layers)
        self.save = [4, 6, 9, 12, 15, 18, 21]
        self._engines: Dict = {}
        self.register_load_state_dict_post_hook(lambda m, _k: m._engines.clear())

    def load_reference(self, sd):
        """Load a reference `TrackingModel.state_dict()` (same keys) and return self."""
        super().load_state_dict(sd, strict=True)
        return self

    def _engine(self, x):
        from .engine import TrackEngine
        B, _, H, W = x.shape
        u8 = x.dtype == torch.uint8
        if u8:
            B, H, W, _ = x.shape
        dt = torch.bfloat16 if x.dtype == torch.bfloat16 else torch.float32
        key = (B, H, W, x.dtype, str(x.device))
        if key not in self._engines:
            self._engines[key] = TrackEngine(self.arch, self.state_dict(), H, W, batch=B, dtype=dt, device=x.device,
                                             input_format="u8" if u8 else "f32", scale_boxes=False)
        return self._engines[key]

    def predict(self, x, is_first=False, profile=False, visualize=False, batch=None, augment=False):
        ops._need_gpu(x)
        eng = self._engine(x)
        out = eng.forward(x if x.dtype == torch.uint8 else x.float())
        return _reference_outputs(eng, out, self.model[-1])

    def forward(self, x, *args, **kwargs):
        return self.predict(x, *args, **kwargs)

    @torch.no_grad()
    def inference_single_image(self, img, ori_img_size, track_instances=None, track_slots: int = 100):
        """Upstream MOTR's per-frame entry (MOTR/models/motr.py:580-598) over the temporal engine:
        img float [1,3,H,W] in [0,1] (H, W multiples of 32); `track_instances=None` starts a sequence (`_generate_empty_tracks`
        there, `reset_sequence` here), otherwise pass the Instances returned by the previous call -- the query memory itself
        lives in HBM and is updated in place, the Instances is the caller's view of it (and the continuity token).
        Returns {'track_instances': live tracks after the ID lifecycle + QIM update, with `boxes` (xyxy in `ori_img_size`
        pixels, TrackerPostProcess motr.py:225-246), `scores`, `labels`, `obj_idxes`, `pred_boxes`, `pred_logits`,
        `output_embedding`, `query_pos`, `ref_pts`; 'ref_pts': the decoder reference points in pixels}."""
        from .engine import TrackEngine
        ops._need_gpu(img)
        if img.dim() != 4 or img.shape[0] != 1 or img.shape[1] != 3:
            raise ValueError("img must be [1, 3, H, W]")
        _, _, H, W = img.shape
        key = ("temporal", H, W, str(img.device), int(track_slots))
        if key not in self._engines:
            self._engines[key] = TrackEngine(self.arch, self.state_dict(), H, W, batch=1, dtype=torch.float32, device=img.device,
                                             input_format="f32", scale_boxes=False, temporal=int(track_slots))
            self._temporal_token = None
        eng = self._engines[key]
        if track_instances is None:
            eng.reset_sequence()
        elif track_instances is not getattr(self, "_temporal_token", None):
            raise ValueError("track_instances must be None (new sequence) or the Instances returned by the previous call: the "
                             "query memory is device-resident state of this model")
        out = eng.forward(img.float())
        n = int(out["n_tracks"][0])
        img_h, img_w = ori_img_size
        scale = torch.tensor([img_w, img_h, img_w, img_h], dtype=torch.float32, device=img.device)
        live = (out["obj_idxes"][0] >= 0).nonzero().view(-1)          # rows of this frame that are tracks now, in query order
        inst = Instances((img_h, img_w))
        pb = out["boxes"][0][live].clone()
        inst.pred_boxes = pb
        inst.boxes = torch.cat([pb[:, :2] - pb[:, 2:] / 2, pb[:, :2] + pb[:, 2:] / 2], -1) * scale
        lg = out["logits"][0][live].clone()
        inst.pred_logits = lg
        inst.scores = out["scores"][0][live].clone()
        inst.labels = lg.argmax(-1)
        inst.obj_idxes = out["obj_idxes"][0][live].clone()
        inst.output_embedding = out["hs"][0][live].clone()
        inst.query_pos = out["trk_qpos"][0][:n].clone()                # memory slots (compacted in query order): the QIM-updated embedding
        inst.ref_pts = out["trk_ref"][0][:n].clone()
        self._temporal_token = inst
        ref = ops.sigmoid_f32(eng.refer_all.view(-1, 4).contiguous())[:, :2] * scale[:2]
        return {"track_instances": inst, "ref_pts": ref.clone()}

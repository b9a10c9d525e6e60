"""TrackEngine: the per-frame tracking step as a static launch plan over libmoyolo.so.

One engine = one (architecture, weights, B frames, H x W, dtype) plan: every activation buffer is
allocated once (channels-last, concat buffers shared by their producers), every weight is
re-laid-out once ([N, Kpad] rows, BN folded to fp32 scale/shift), and the step is a fixed list of
C-ABI launches with pre-built argument blocks.  `forward` only enqueues that list on the current
HIP stream (no host sync, no allocation), so it can be captured into a hipGraph (`capture`).

Reference path restated: preprocess a1 (fused into the stem), backbone/neck a3-a6,
MYDecoder a8-a14, per-frame ID assignment + predictor rows a15/a16/a20.
PyTorch is used for device memory and streams only.
"""
from __future__ import annotations

import dataclasses
import os

import ctypes as C
from typing import Dict, List

import torch

from . import _lib as L
from .config import TrackArch, level_shapes

BN_EPS = 1e-3   # ultralytics/utils/torch_utils.py:262


@dataclasses.dataclass(frozen=True)
class PlanOptions:
    """How a TrackEngine lays its launch plan out -- every A/B switch of the plan as an explicit constructor argument (round 6: rounds
    1-5 read 18 `MOY_*` environment variables here; process-wide state a host could not see).  The defaults ARE the shipped plan; the
    product path never reads the environment.  `PlanOptions.from_env()` maps the old variable names onto the fields for the lab
    scripts under tools/ (A/B runs), `bench.py --plan key=value,...` builds one from its argument."""
    stem_x3: bool = True          # round 6, f32x3 only: the stem on the matrix cores with split weights (bytes are exact fp16 values)
    fuse_stem: bool = True        # preprocess + layer 0 + layer 1 as one launch (16-bit, uint8 frames)
    fuse_upsample: bool = True    # Upsample + Concat folded into the consuming C2f's cv1
    fuse_c2f: bool = True         # the first C2f block as one launch
    value_planes: int = 2         # 2: head planes [layer][head][token][32]; 1: one [B*S, 256] matrix per layer; 0: one [B*S, 6*256] matrix
    fold_proj: bool = True        # input_proj folded into value projection / score pass (where the library has the launches)
    fold_min_rows: int = 8192     # 16-bit engines fold from this many rows on the smallest level (= where the library admits the value / score launches; round 6: was 65536)
    score_runs: bool = True       # score pass over the valid tokens only
    p3_raw: bool = True           # level 0 of the deformable attention gathered raw and projected after the bilinear sum
    query_order: bool = False     # the gather walks a frame's queries in Morton order of their reference points (round 6: L1 hit rate 72.9 ->
                                  # 76.9 %, kernel 1-2 % SLOWER -- it is bound by the texture addresser, not by misses: off)
    mlp_head: bool = True         # the three-layer box heads as one launch
    dec_mid: bool = True          # out_proj + norm1 + offsets | weights linear as one launch
    dec_tail: bool = True         # output_proj + norm2 + FFN + norm3 + box refinement as one launch
    qkv_split: int = 1            # q | k and v as two plain products over the tail's x + query_pos: 0 never, 1 at bench scale, 2 always
    qkv_fuse: bool = False        # the tail projects the next layer's q | k | v (measured neutral)
    w_packed: bool = True         # weights of the row-wise decoder kernels in MFMA-fragment order
    post_1x1: bool = True         # the cv1 of the second C2f inside the down-sampling conv that feeds it
    fork_value: int = 0           # compute units of a forked P3 value launch (measured: no gain; 0 = off)
    fork_small_value: int = 0     # round 6: below this many value rows (B x S) the value projection of the classic plan runs on a side stream
                                  # beside the query selection / first self-attention.  MEASURED at four frames per step: 1.258 ms forked
                                  # against 1.224 ms in plan order (the value launch fills the chip and only takes turns with the chain): off
    qkv_fuse_small: bool = True   # round 6: below 65536 decoder rows the tail projects the next layer's q | k | v (a launch less per layer)
    assume_wreg: bool = True      # the library's weight-stationary kernel may be planned for (False with a lab library run under MOY_GEMM_WREG=0)

    _ENV = dict(fuse_stem="MOY_FUSE_STEM", fuse_upsample="MOY_FUSE_UPSAMPLE", fuse_c2f="MOY_FUSE_C2F", value_planes="MOY_VALUE_PLANES",
                stem_x3="MOY_STEM_X3", fold_proj="MOY_FOLD_PROJ", fold_min_rows="MOY_FOLD_MIN_ROWS", score_runs="MOY_SCORE_RUNS", p3_raw="MOY_P3_RAW", query_order="MOY_Q_ORDER", mlp_head="MOY_MLP_HEAD",
                dec_mid="MOY_DEC_MID", dec_tail="MOY_DEC_TAIL", qkv_split="MOY_QKV_SPLIT", qkv_fuse="MOY_QKV_FUSE", w_packed="MOY_W_PACKED",
                post_1x1="MOY_POST_1X1", fork_value="MOY_FORK_VALUE", assume_wreg="MOY_GEMM_WREG", fork_small_value="MOY_FORK_SMALL",
                qkv_fuse_small="MOY_QKV_FUSE_SMALL")

    @classmethod
    def parse(cls, text: str) -> "PlanOptions":
        """'p3_raw=0,qkv_fuse=1' -> PlanOptions (unknown keys raise)."""
        kw = {}
        fields = {f.name: f for f in dataclasses.fields(cls)}
        for item in filter(None, (t.strip() for t in (text or "").split(","))):
            k, _, v = item.partition("=")
            if k not in fields:
                raise ValueError(f"unknown plan option {k!r}; known: {sorted(fields)}")
            kw[k] = (v.strip() not in ("0", "", "false", "False")) if fields[k].type in (bool, "bool") else int(v)
        return cls(**kw)

    @classmethod
    def from_env(cls, env=None) -> "PlanOptions":
        """LAB ONLY: the rounds 1-5 variable names (MOY_FOLD_PROJ=0, ...) onto the fields."""
        env = os.environ if env is None else env
        items = [f"{k}={env[v]}" for k, v in cls._ENV.items() if v in env]
        return cls.parse(",".join(items))


class View:
    """A [rows, cols] window of a row-major 2-D device buffer (channel slice of a concat buffer)."""

    def __init__(self, buf: torch.Tensor, off: int = 0, cols: int | None = None):
        assert buf.dim() == 2 and buf.is_contiguous()
        self.buf, self.off = buf, off
        self.cols = buf.shape[1] - off if cols is None else cols
        assert 0 <= off and off + self.cols <= buf.shape[1]

    @property
    def ptr(self):
        return self.buf.data_ptr() + self.off * self.buf.element_size()

    @property
    def ld(self):
        return self.buf.shape[1]

    @property
    def rows(self):
        return self.buf.shape[0]

    def slice(self, off, cols):
        return View(self.buf, self.off + off, cols)

    def tensor(self):
        return self.buf[:, self.off:self.off + self.cols]


def _code(dtype):
    return {torch.float32: L.F32, torch.bfloat16: L.BF16, torch.float16: L.F16}[dtype]


class TrackEngine:
    def __init__(self, arch: TrackArch, state_dict: Dict[str, torch.Tensor], H: int, W: int, batch: int = 1,
                 dtype: torch.dtype = torch.float32, device="cuda", input_format: str = "u8", conf: float = 0.25,
                 score_thresh: float = 0.4, scale_boxes: bool = True, head_only: bool = False,
                 level_shapes_override=None, side_state: bool = False, iou: float = 0.7, max_det: int = 300, orig_hw=None,
                 temporal: int = 0, filter_score_thresh: float = 0.5, miss_tolerance: int = 5, n_inputs: int = 1,
                 split_f16: bool = False, track_content: str = "decoder_output", options: PlanOptions | None = None):
        if not torch.cuda.is_available():
            raise L.MoyoloError("TrackEngine needs a HIP device (no CPU path)")
        self.lib = L.lib()
        self.opt = options if options is not None else PlanOptions()
        self.arch, self.H, self.W, self.B = arch, H, W, batch
        self.dtype, self.code, self.dev = dtype, _code(dtype), torch.device(device)
        # split_f16 (fp32 engines, round 5): every product of `moy_gemm` on the 16-bit matrix cores in split precision (MOY_F32X3:
        # x = hi + lo * 2^-11, three fp16 products, fp32 accumulation -- about 22 mantissa bits at 16/3 of the fp32 matrix rate);
        # tensors, the other kernels and the plan stay those of the fp32 engine
        if split_f16 and dtype != torch.float32:
            raise ValueError("split_f16 is a mode of the fp32 engine")
        self.split_f16 = bool(split_f16)
        self.code_gemm = L.F32X3 if split_f16 else self.code
        self._esz = 4 if dtype == torch.float32 else 2
        self.input_format = input_format
        self.conf, self.score_thresh = conf, score_thresh
        # rows are scaled to the ORIGINAL frame (predict.py:61-76); it differs from (H, W) when the predictor stretch-resized
        oh, ow = orig_hw if (orig_hw is not None and arch.head_kind != "detect") else (H, W)
        self.img_wh = (float(ow), float(oh)) if scale_boxes else (1.0, 1.0)
        self.head_only = head_only
        self.iou, self.max_det, self.orig_hw = iou, max_det, orig_hw      # Detect head (config C1) only
        self.side_state = side_state     # keep the output-invisible tracker copy + FSQM memory on device
        # temporal mode: `temporal` = track slots per sequence; batch element b is then
        # SEQUENCE b (frames of one sequence are dependent), its query memory persists across forward() calls
        self.n_max = int(temporal)
        self.filter_score_thresh, self.miss_tolerance = filter_score_thresh, miss_tolerance
        if self.n_max and (side_state or head_only):
            raise ValueError("temporal mode excludes side_state / head_only")
        # temporal mode, CONTENT embedding of a carried track: "decoder_output" = its own decoder output of the
        # previous frame (upstream MOTR, motr.py:545-577: the default); "class_embed" = the fork's own visible design,
        # denoising_class_embed.weight[argmax of the track's class scores] (head.py:888-900, 917-919, 1109-1110)
        if track_content not in ("decoder_output", "class_embed"):
            raise ValueError(track_content)
        self.track_content = track_content
        self.shapes = [tuple(s) for s in level_shapes_override] if level_shapes_override else level_shapes(H, W)
        self.S = sum(h * w for h, w in self.shapes)
        self._keep: List[torch.Tensor] = []          # device tensors referenced by raw pointers
        self._act_bufs: List[torch.Tensor] = []      # the subset of _keep that `_buf` handed out as per-step activations
        self._value_launches: List = []              # (GemmArgs, rows per frame) of the value-projection launches (stress.value_planes_vs_tiled)
        self._steps: List = []                       # (fn, args tuple) launches
        self.meta: List[dict] = []                   # per launch: name, algorithmic bytes, flops
        self.sd = {k: v.detach().float().cpu() for k, v in state_dict.items()}
        self._graph = None
        # n_inputs static input buffers ("slots"): the producer of the frames (decoder / resize stage / test) writes slot i while
        # the engine still reads slot j; every slot has its own captured graph, so a step never copies frames (round 1 copied
        # B x H x W x 3 bytes into one static input per step)
        self.n_inputs = max(1, int(n_inputs))
        self._graphs: Dict[int, object] = {}
        with torch.no_grad():
            self._build()
        self._plan_fork()

    # ------------------------------------------------------------------ helpers
    def _dev(self, t, dtype=None):
        t = t.to(device=self.dev, dtype=dtype or t.dtype).contiguous()
        self._keep.append(t)
        return t

    def _buf(self, rows, cols, dtype=None, static=False):
        """An activation buffer of the plan.  static=True: the buffer carries state the step does not rewrite (constants left at plan
        build, the temporal query memory) -- `poison_activations` leaves those alone."""
        t = torch.zeros(rows, cols, device=self.dev, dtype=dtype or self.dtype)
        self._keep.append(t)
        if not static:
            self._act_bufs.append(t)
        return t

    def poison_activations(self):
        """Self-check hook (mo_yolo_amd/stress.py): fill every floating-point activation buffer the step is supposed to (re)write with
        NaN, so that a pass that reads something it never wrote shows it in its outputs.  Returns the number of buffers filled."""
        n = 0
        for t in self._act_bufs:
            if t.is_floating_point():
                t.fill_(float("nan"))
                n += 1
        return n

    def _kpad(self, K):
        bk = 32 if self.dtype == torch.float32 else 64
        return (K + bk - 1) // bk * bk

    def _weight(self, w2d):
        """[N, K] fp32 -> device [N, Kpad] in the engine dtype (zero padded)."""
        N, K = w2d.shape
        out = torch.zeros(N, self._kpad(K), dtype=torch.float32)
        out[:, :K] = w2d
        if self.split_f16:      # MOY_F32X3 takes its weights pre-split: fp16 [2][N][Kpad], heads and scaled remainders
            hi = out.to(torch.float16)
            return self._dev(torch.stack([hi, ((out - hi.float()) * 2048.0).to(torch.float16)]))
        return self._dev(out, self.dtype)

    def _bn(self, p):
        sd = self.sd
        scale = sd[p + ".weight"] / torch.sqrt(sd[p + ".running_var"] + BN_EPS)
        shift = sd[p + ".bias"] - sd[p + ".running_mean"] * scale
        return self._dev(scale), self._dev(shift)

    def _add(self, fn, *args, meta=None):
        self._steps.append((fn, args))
        m = dict(name=fn.__name__, bytes=0, flops=0)
        if meta:
            m.update(meta)
        self.meta.append(m)

    def _gemm(self, A: View, Wt, N, K, C_: View, M, *, ksize=1, stride=1, geom=None, scale=None, shift=None, act=0,
              A2: View | None = None, a_rows=None, a_mask=None, mask_period=0, R: View | None = None, ln=None,
              out_f32=False, c_rpb=0, c_bstride=0, dot=None, pre=None, a2_cols=0, planes=None, runs=None, meta_scale=1.0, post=None):
        a = L.GemmArgs()
        a.A, a.lda = A.ptr, A.ld
        a.A2 = A2.ptr if A2 is not None else None
        if A2 is not None:
            assert A2.ld == A.ld
        a.a_rows = a_rows.data_ptr() if a_rows is not None else None
        a.a_rows_bound = A.rows if a_rows is not None else 0
        a.a_mask = a_mask.data_ptr() if a_mask is not None else None
        a.mask_period = mask_period
        a.W, a.M, a.N, a.K = Wt.data_ptr(), M, N, K
        a.ksize, a.stride = ksize, stride
        if geom is not None:
            a.B, a.Hin, a.Win, a.Hout, a.Wout, a.Cin = geom
        a.scale = scale.data_ptr() if scale is not None else None
        a.shift = shift.data_ptr() if shift is not None else None
        a.act = act
        if R is not None:
            a.R, a.ldr = R.ptr, R.ld
        if ln is not None:
            a.ln_g, a.ln_b = ln[0].data_ptr(), ln[1].data_ptr()
        if C_ is not None:
            a.C, a.ldc = C_.ptr, C_.ld
        a.out_f32, a.dtype = int(out_f32), self.code_gemm
        a.c_rows_per_batch, a.c_batch_stride = c_rpb, c_bstride
        if dot is not None:
            a.dot_w, a.dot_b, a.dot_out, a.dot_n = dot[0].data_ptr(), dot[1].data_ptr(), dot[2].data_ptr(), dot[0].shape[0]
        a.a2_cols = a2_cols
        if planes is not None:
            a.plane_cols, a.plane_stride = planes
        if pre is not None:
            pt, ph, pw = pre
            a.pre, a.ld_pre, a.pre_h, a.pre_w = pt.data_ptr(), pt.stride(0), ph, pw
        if runs is not None:    # score pass over the valid rectangle of ONE level read from that level's own tensor (moy_gemm_args.run_*)
            a.run_levels, a.run_period = 1, runs["period"]
            a.run_tok0[0], a.run_pitch[0], a.run_len[0], a.run_rows[0] = runs["tok0"], runs["pitch"], runs["len"], runs["rows"]
            a.run_a_period, a.run_a_off = runs["a_period"], runs["a_off"]
        n_out = N
        if post is not None:    # (W2, scale2, shift2, n2, act2): the conv's only consumer, a 1x1 conv, applied to the finished tiles on chip
            a.post_W, a.post_scale, a.post_shift, a.post_n, a.post_act = post[0].data_ptr(), post[1].data_ptr(), post[2].data_ptr(), post[3], post[4]
            n_out = post[3]
        self._keep.append(a)
        esz = 4 if self.dtype == torch.float32 else 2
        if geom is not None:
            a_elems = geom[0] * geom[1] * geom[2] * geom[5]      # every input pixel read once
        else:
            a_elems = M * K * (2 if A2 is not None else 1)
        alg = (a_elems + N * K) * esz + (M * n_out * (4 if out_f32 else esz) if C_ is not None else 0) + (M * N * esz if R is not None else 0)
        # (a name = one kernel template as `rocprofv3 --stats` lists it: bench.py ranks kernels by the total time of a NAME; the 3x3 convs with
        #  the bottleneck shortcut are a template of their own)
        tag = (f"gemm{ksize}x{ksize}" + ("s2" if stride == 2 else "") + ("+ln" if ln is not None else "") + ("+1x1" if post is not None else "")
               + ("+res" if R is not None and ksize == 3 else ""))
        fl = 2 * M * N * K + (2 * M * N * n_out if post is not None else 0)
        if post is not None:
            alg += N * n_out * esz
        self._add(self.lib.moy_gemm, C.byref(a), meta=dict(name=f"{tag} M{M} N{N} K{K}", bytes=int(alg * meta_scale), flops=int(fl * meta_scale)))

    # conv + BN + SiLU on channels-last views
    def _conv(self, p, x: View, hw_in, cin, cout, k, s, out: View, R: View | None = None, act=L.ACT_SILU, up_src=None, post=None):
        """post = prefix of a 1x1 Conv that is this 3x3 conv's only consumer: folded into the launch (`out` then takes ITS output);
        returns False -- nothing planned -- when the library has no fused form for the shape / launch size (asked on the host:
        `moy_gemm_query`, nothing is launched at plan build).
        up_src = (u view, (h, w), cu): the conv's input is Concat[Upsample2x(u), x] (u at half resolution, cu channels first)."""
        sd = self.sd
        w = sd[p + ".conv.weight"]
        scale, shift = self._bn(p + ".bn")
        Hin, Win = hw_in
        if k == 1 and up_src is not None:
            # Conv1x1(Concat[Upsample(u), x]) = Upsample(W_u . u) + W_x . x: the u half of the product is formed at HALF
            # resolution (fp32, no rounding) and seeds the accumulators of the full-resolution GEMM over x alone
            u, (hu, wu), cu = up_src
            assert (2 * hu, 2 * wu) == (Hin, Win) and w.shape[1] == cu + cin
            w2 = w.reshape(cout, cu + cin)
            t = self._buf(self.B * hu * wu, cout, torch.float32)
            self._gemm(u, self._weight(w2[:, :cu]), cout, cu, View(t), self.B * hu * wu, out_f32=True)
            self._gemm(x, self._weight(w2[:, cu:]), cout, cin, out, self.B * Hin * Win, scale=scale, shift=shift, act=act, R=R,
                       pre=(t, Hin, Win))
            return hw_in
        if k == 1:
            Wt = self._weight(w.reshape(cout, cin))
            self._gemm(x, Wt, cout, cin, out, self.B * Hin * Win, scale=scale, shift=shift, act=act, R=R)
            return hw_in
        Hout, Wout = (Hin + 2 - 3) // s + 1, (Win + 2 - 3) // s + 1
        Wt = self._weight(w.permute(0, 2, 3, 1).reshape(cout, 9 * cin))
        if post is not None:
            w2 = sd[post + ".conv.weight"]
            n2 = w2.shape[0]
            s2_, h2_ = self._bn(post + ".bn")
            self._gemm(x, Wt, cout, 9 * cin, out, self.B * Hout * Wout, ksize=3, stride=s, geom=(self.B, Hin, Win, Hout, Wout, cin),
                       scale=scale, shift=shift, act=act, post=(self._weight(w2.reshape(n2, cout)), s2_, h2_, n2, L.ACT_SILU))
            rc, _ = self._query_last()
            if rc == L.ENOSYS:                     # no fused form for this shape / launch size
                self._steps.pop()
                self.meta.pop()
                return False
            L.check(rc, "moy_gemm_query (conv + folded 1x1 consumer)")
            return (Hout, Wout)
        self._gemm(x, Wt, cout, 9 * cin, out, self.B * Hout * Wout, ksize=3, stride=s,
                   geom=(self.B, Hin, Win, Hout, Wout, cin), scale=scale, shift=shift, act=act, R=R)
        return (Hout, Wout)

    def _query_last(self):
        """(rc, kernel family) the library answers for the `moy_gemm` launch planned last -- validation and kernel eligibility of a
        real call, on the host, without launching (include/moyolo.h: moy_gemm_query)."""
        fn, args = self._steps[-1]
        assert fn is self.lib.moy_gemm or fn.__name__ == "moy_gemm"
        k = C.c_int(0)
        rc = self.lib.moy_gemm_query(args[0], C.byref(k))
        return rc, k.value

    def _linear_w(self, p, rows=None):
        w, b = self.sd[p + ".weight"], self.sd[p + ".bias"]
        if rows is not None:
            w, b = w[rows], b[rows]
        return self._weight(w), self._dev(b)

    def _fuse_stem_l1(self, consumers):
        """Layers 0 and 1 as one launch: uint8 frames, 16-bit engine, Conv(3->32, s2) feeding ONLY Conv(32->64, 3x3, s2)."""
        a = self.arch.layers
        if not self.opt.fuse_stem or self.dtype == torch.float32 or self.input_format != "u8" or len(a) < 2:
            return False
        return (a[0].kind == "Conv" and a[0].k == 3 and a[0].s == 2 and a[0].c1 == 3 and a[0].c2 == 32 and a[1].kind == "Conv"
                and a[1].k == 3 and a[1].s == 2 and a[1].c1 == 32 and a[1].c2 == 64 and list(a[1].src) in ([-1], [0])
                and consumers.get(0, []) == [1] and self.H % 4 == 0 and self.W % 4 == 0)

    # ------------------------------------------------------------------ plan
    def _build(self):
        arch, B, H, W, sd, lib, code = self.arch, self.B, self.H, self.W, self.sd, self.lib, self.code
        nlayers = len(arch.layers)
        outv: Dict[int, View] = {}
        hw: Dict[int, tuple] = {}
        if self.head_only:
            # MOTRTrack.forward surface: the three pyramid levels are the inputs (head.py:191)
            self.input = None
            self.head_inputs = []
            for li, (h_, w_) in enumerate(self.shapes):
                v = View(self._buf(B * h_ * w_, arch.head_ch[li]))
                self.head_inputs.append(v)
            head_src = [(v, hw_) for v, hw_ in zip(self.head_inputs, self.shapes)]
        else:
            if self.input_format == "u8":
                self.inputs = [torch.zeros(B, H, W, 3, device=self.dev, dtype=torch.uint8) for _ in range(self.n_inputs)]
            elif self.input_format == "f32":
                self.inputs = [torch.zeros(B, 3, H, W, device=self.dev, dtype=torch.float32) for _ in range(self.n_inputs)]
            else:
                raise ValueError(self.input_format)
            self.input = self.inputs[0]

            # spatial size per layer output
            cur = (H, W)
            for Ls in arch.layers:
                src_hw = hw[Ls.src[0]] if Ls.src[0] >= 0 else (H, W)
                if Ls.kind == "Conv":
                    cur = ((src_hw[0] + 2 * (Ls.k // 2) - Ls.k) // Ls.s + 1, (src_hw[1] + 2 * (Ls.k // 2) - Ls.k) // Ls.s + 1)
                elif Ls.kind == "Upsample":
                    cur = (src_hw[0] * 2, src_hw[1] * 2)
                else:
                    cur = src_hw
                hw[Ls.i] = cur

            # homes: outputs that feed a Concat live inside the concat buffer
            home: Dict[int, View] = {}
            for Ls in arch.layers:
                if Ls.kind == "Concat":
                    h_, w_ = hw[Ls.i]
                    buf = self._buf(B * h_ * w_, Ls.c2)
                    off = 0
                    for j in Ls.src:
                        cj = arch.layers[j].c2
                        assert j not in home, "a tensor may live in one concat buffer only"
                        home[j] = View(buf, off, cj)
                        off += cj
                    home[Ls.i] = View(buf)
            def out_view(i, c):
                if i in home:
                    return home[i]
                h_, w_ = hw[i]
                return View(self._buf(B * h_ * w_, c))

            # Upsample -> Concat -> C2f (yolo_track.yaml:28-33): when the concat's only consumer is that C2f, neither the
            # upsampled tensor nor its copy into the concat buffer is ever formed (see _conv up_src)
            consumers: Dict[int, list] = {}
            for Ls in arch.layers:
                for j in Ls.src:
                    consumers.setdefault(j if j >= 0 else Ls.i - 1, []).append(Ls.i)
            virt_up: Dict[int, tuple] = {}       # Upsample layer -> (source view, source hw, channels)
            virt_cat: Dict[int, tuple] = {}      # Concat layer -> (u view, u hw, cu, x view, cx)
            def fusable(cat):
                if cat.kind != "Concat" or len(cat.src) != 2:
                    return False
                up = arch.layers[cat.src[0]]
                cons = [arch.layers[c] for c in consumers.get(cat.i, [])]
                return (up.kind == "Upsample" and consumers.get(up.i, []) == [cat.i] and len(cons) == 1 and cons[0].kind == "C2f"
                        and self.opt.fuse_upsample)
            self._layer_first_step: Dict[int, int] = {}
            pending_post: Dict[int, tuple] = {}
            self.post_fused_layers = set()
            for Ls in arch.layers:
                p = f"model.{Ls.i}"
                self._layer_first_step[Ls.i] = len(self._steps)
                x = outv[Ls.src[0]] if Ls.src[0] >= 0 else None
                hin = hw[Ls.src[0]] if Ls.src[0] >= 0 else (H, W)
                if Ls.kind == "Conv" and Ls.i == 0 and self._fuse_stem_l1(consumers):
                    # layers 0 + 1 in one launch (csrc/stem_l1.hip): the 32-channel half-resolution tensor never reaches HBM
                    from .ops import stem_weights_fused
                    L1 = arch.layers[1]
                    o1 = out_view(1, L1.c2)
                    s0, h0 = self._bn("model.0.bn")
                    s1, h1 = self._bn("model.1.bn")
                    w0 = self._dev(stem_weights_fused(sd["model.0.conv.weight"], self.dtype))
                    w1 = self._weight(sd["model.1.conv.weight"].permute(0, 2, 3, 1).reshape(L1.c2, 9 * L1.c1))
                    ho, wo = H // 4, W // 4
                    self._add(lib.moy_stem_l1_fused, self.input.data_ptr(), B, H, W, w0.data_ptr(), s0.data_ptr(), h0.data_ptr(), w1.data_ptr(),
                              s1.data_ptr(), h1.data_ptr(), o1.ptr, o1.ld, code,
                              meta=dict(name=f"stem+conv1 fused M{B * ho * wo} N{L1.c2}", bytes=B * H * W * 3 + B * ho * wo * L1.c2 * 2,
                                        flops=2 * B * (H // 2) * (W // 2) * 32 * 27 + 2 * B * ho * wo * L1.c2 * 288))
                    self._stem_step = len(self._steps) - 1
                    outv[0], outv[1] = None, o1
                    self._fused_l1 = True
                elif Ls.kind == "Conv" and Ls.i == 1 and getattr(self, "_fused_l1", False):
                    pass                                   # produced by the fused launch above
                elif Ls.kind == "Conv" and Ls.i == 0:
                    assert Ls.k == 3 and Ls.s == 2 and Ls.c1 == 3
                    o = out_view(0, Ls.c2)
                    scale, shift = self._bn(p + ".bn")
                    if self.dtype == torch.bfloat16 and self.input_format == "u8" and Ls.c2 in (16, 32, 64):
                        from .ops import stem_weights_mfma          # matrix-core stem (K 27 -> 32)
                        wpad = self._dev(stem_weights_mfma(sd[p + ".conv.weight"]))
                        self._add(lib.moy_stem_conv_mfma, self.input.data_ptr(), B, H, W, wpad.data_ptr(), scale.data_ptr(),
                                  shift.data_ptr(), Ls.c2, o.ptr, o.ld,
                                  meta=dict(name=f"stem_mfma M{B * (H // 2) * (W // 2)} N{Ls.c2}", bytes=B * H * W * 3 + B * (H // 2) * (W // 2) * Ls.c2 * 2,
                                            flops=2 * B * (H // 2) * (W // 2) * Ls.c2 * 27))
                    elif (self.split_f16 and self.opt.stem_x3 and self.input_format == "u8" and Ls.c2 in (16, 32, 64) and o.ld % 4 == 0
                          and W % 4 == 0 and B * H * W * 3 < 2 ** 32 - 4):
                        # round 6: the pixel bytes are exact fp16 values -> two products per tile with the split weights (moy_stem_conv_x3)
                        from .ops import stem_weights_x3
                        wsp = self._dev(stem_weights_x3(sd[p + ".conv.weight"]))
                        self._add(lib.moy_stem_conv_x3, self.input.data_ptr(), B, H, W, wsp.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                  Ls.c2, o.ptr, o.ld,
                                  meta=dict(name=f"stem_x3 M{B * (H // 2) * (W // 2)} N{Ls.c2}", bytes=B * H * W * 3 + B * (H // 2) * (W // 2) * Ls.c2 * 4,
                                            flops=2 * B * (H // 2) * (W // 2) * Ls.c2 * 27))
                    else:
                        wst = self._dev(sd[p + ".conv.weight"].permute(2, 3, 1, 0).reshape(27, Ls.c2))
                        esz = 4 if self.dtype == torch.float32 else 2
                        self._add(lib.moy_stem_conv, self.input.data_ptr(), 0 if self.input_format == "u8" else 1, B, H, W,
                                  wst.data_ptr(), scale.data_ptr(), shift.data_ptr(), Ls.c2, o.ptr, o.ld, code,
                                  meta=dict(name=f"stem M{B * (H // 2) * (W // 2)} N{Ls.c2}",
                                            bytes=B * H * W * 3 * (1 if self.input_format == "u8" else 4) + B * (H // 2) * (W // 2) * Ls.c2 * esz,
                                            flops=2 * B * (H // 2) * (W // 2) * Ls.c2 * 27))
                    self._stem_step = len(self._steps) - 1       # its first argument is the input slot's pointer
                    outv[0] = o
                elif Ls.kind == "Conv" and self._post_candidate(Ls, consumers, home):
                    # its only consumer is the cv1 of the next C2f: planned there, as ONE launch when the library has the fused form
                    pending_post[Ls.i] = (p, x, hin, Ls)
                    outv[Ls.i] = None
                elif Ls.kind == "Conv":
                    o = out_view(Ls.i, Ls.c2)
                    self._conv(p, x, hin, Ls.c1, Ls.c2, Ls.k, Ls.s, o)
                    outv[Ls.i] = o
                elif (Ls.kind == "C2f" and Ls.c1 == 64 and Ls.c2 == 64 and Ls.n == 1 and Ls.shortcut and Ls.src[0] not in virt_cat
                      and self.dtype != torch.float32 and self.opt.fuse_c2f
                      and hin[0] * hin[1] * 64 * 2 <= 0x3fffffff):
                    # the whole block in one launch (csrc/c2f_fused.hip): y0 | y1, z and y2 never reach HBM
                    c, (h_, w_) = 32, hin
                    a = L.C2fArgs()
                    o = out_view(Ls.i, Ls.c2)
                    a.x, a.ldx, a.B, a.H, a.W = x.ptr, x.ld, B, h_, w_
                    wt = {}
                    for key, q_, kk in (("cv1", p + ".cv1", 1), ("m1", p + ".m.0.cv1", 3), ("m2", p + ".m.0.cv2", 3), ("cv2", p + ".cv2", 1)):
                        w = sd[q_ + ".conv.weight"]
                        w2 = w.reshape(w.shape[0], -1) if kk == 1 else w.permute(0, 2, 3, 1).reshape(w.shape[0], -1)
                        wt[key] = (self._weight(w2),) + self._bn(q_ + ".bn")
                    a.w_cv1, a.kp_cv1, a.scale_cv1, a.shift_cv1 = wt["cv1"][0].data_ptr(), wt["cv1"][0].shape[1], wt["cv1"][1].data_ptr(), wt["cv1"][2].data_ptr()
                    a.w_m1, a.scale_m1, a.shift_m1 = wt["m1"][0].data_ptr(), wt["m1"][1].data_ptr(), wt["m1"][2].data_ptr()
                    a.w_m2, a.scale_m2, a.shift_m2, a.kp_m = wt["m2"][0].data_ptr(), wt["m2"][1].data_ptr(), wt["m2"][2].data_ptr(), wt["m1"][0].shape[1]
                    a.w_cv2, a.kp_cv2, a.scale_cv2, a.shift_cv2 = wt["cv2"][0].data_ptr(), wt["cv2"][0].shape[1], wt["cv2"][1].data_ptr(), wt["cv2"][2].data_ptr()
                    a.out, a.ldo, a.dtype = o.ptr, o.ld, code
                    self._keep.append(a)
                    M_ = B * h_ * w_
                    self._add(lib.moy_c2f_fused, C.byref(a), meta=dict(
                        name=f"c2f fused M{M_} 64->[32|32]->64", bytes=M_ * 128 * self._esz,
                        flops=2 * M_ * (64 * 64 + 2 * 32 * 288 + 96 * 64)))
                    outv[Ls.i] = o
                elif Ls.kind == "C2f":
                    c = Ls.c2 // 2
                    h_, w_ = hin
                    cat = View(self._buf(B * h_ * w_, (2 + Ls.n) * c))
                    tmp = View(self._buf(B * h_ * w_, c))
                    fused_cv1 = False
                    if Ls.src[0] in pending_post:
                        pp, px, phin, PL = pending_post.pop(Ls.src[0])
                        fused_cv1 = bool(self._conv(pp, px, phin, PL.c1, PL.c2, PL.k, PL.s, cat.slice(0, 2 * c), post=p + ".cv1"))
                        if fused_cv1:
                            self.post_fused_layers.add(PL.i)       # never materialised: its tiles only ever exist in LDS
                        else:
                            x = out_view(PL.i, PL.c2)
                            self._conv(pp, px, phin, PL.c1, PL.c2, PL.k, PL.s, x)
                            outv[PL.i] = x
                    if fused_cv1:
                        pass
                    elif Ls.src[0] in virt_cat:
                        u, uhw, cu, xs, cx = virt_cat[Ls.src[0]]
                        self._conv(p + ".cv1", xs, hin, cx, 2 * c, 1, 1, cat.slice(0, 2 * c), up_src=(u, uhw, cu))
                    else:
                        self._conv(p + ".cv1", x, hin, Ls.c1, 2 * c, 1, 1, cat.slice(0, 2 * c))
                    for j in range(Ls.n):
                        src = cat.slice((1 + j) * c, c)
                        self._conv(f"{p}.m.{j}.cv1", src, hin, c, c, 3, 1, tmp)
                        self._conv(f"{p}.m.{j}.cv2", tmp, hin, c, c, 3, 1, cat.slice((2 + j) * c, c),
                                   R=src if Ls.shortcut else None)
                    o = out_view(Ls.i, Ls.c2)
                    self._conv(p + ".cv2", cat, hin, (2 + Ls.n) * c, Ls.c2, 1, 1, o)
                    outv[Ls.i] = o
                elif Ls.kind == "SPPF":
                    c_ = Ls.c1 // 2
                    h_, w_ = hin
                    cat = View(self._buf(B * h_ * w_, 4 * c_))
                    self._conv(p + ".cv1", x, hin, Ls.c1, c_, 1, 1, cat.slice(0, c_))
                    s0, s1, s2, s3 = (cat.slice(i * c_, c_) for i in range(4))
                    self._add(lib.moy_sppf_pool, s0.ptr, s0.ld, B, h_, w_, c_, s1.ptr, s2.ptr, s3.ptr, cat.ld, code,
                              meta=dict(name=f"sppf_pool M{B * h_ * w_} C{c_}", bytes=4 * B * h_ * w_ * c_ * self._esz, flops=0))
                    o = out_view(Ls.i, Ls.c2)
                    self._conv(p + ".cv2", cat, hin, 4 * c_, Ls.c2, 1, 1, o)
                    outv[Ls.i] = o
                elif Ls.kind == "Upsample":
                    cat = arch.layers[consumers[Ls.i][0]] if len(consumers.get(Ls.i, [])) == 1 else None
                    if cat is not None and fusable(cat):
                        virt_up[Ls.i] = (x, hin, Ls.c1)
                        outv[Ls.i] = None
                    else:
                        o = out_view(Ls.i, Ls.c2)
                        self._add(lib.moy_upsample2x, x.ptr, x.ld, B, hin[0], hin[1], Ls.c1, o.ptr, o.ld, code,
                                  meta=dict(name=f"upsample2x M{B * hin[0] * hin[1]} C{Ls.c1}", bytes=5 * B * hin[0] * hin[1] * Ls.c1 * self._esz, flops=0))
                        outv[Ls.i] = o
                elif Ls.kind == "Concat":
                    outv[Ls.i] = home[Ls.i]
                    if Ls.src[0] in virt_up:
                        u, uhw, cu = virt_up[Ls.src[0]]
                        virt_cat[Ls.i] = (u, uhw, cu, home[Ls.src[1]], arch.layers[Ls.src[1]].c2)
                else:
                    raise ValueError(Ls.kind)
            self.layer_views, self.layer_hw = outv, hw
            assert not pending_post
            self.virtual_layers = set(virt_up) | set(virt_cat) | self.post_fused_layers     # never materialised (Upsample + Concat folded into the C2f; a Conv folded with its 1x1 consumer)

            head_src = [(outv[j], hw[j]) for j in (15, 18, 21)]
        self._head_start = len(self._steps)
        if arch.head_kind == "detect":
            self._build_detect_head(head_src)
            return
        # ---------------- head (MYDecoder)
        d = f"model.{nlayers}.decoder"
        hd, nq, nc, S, nl = arch.hd, arch.nq, arch.nc, self.S, arch.nl
        assert hd == 256 and arch.nh == 8 and arch.ndp == 4, "kernels are specialised to hd 256 / 8 heads / 4 points"
        # anchors / valid mask are input independent: host precompute with the reference formula
        anchors, valid = _generate_anchors(self.shapes)
        self.anchors = self._dev(anchors[0])                        # [S, 4] (+inf at masked tokens)
        self.valid = self._dev(valid[0, :, 0].to(torch.uint8))     # [S]
        for (src_view, src_hw), hw_ in zip(head_src, self.shapes):
            assert tuple(src_hw) == tuple(hw_)
        # Round 4: input_proj (Conv1x1 + BN, NO activation: head.py:838-839) is a linear map between two linear consumers, so at bench
        # scale it is folded into them -- value projection and score pass read each level's own tensor with composed weights, the
        # projected features exist only for the nq selected tokens (`_fold_plan` says when; MOY_FOLD_PROJ=0 keeps the classic plan)
        # (whether level 0 will be sampled raw is known before the fold is asked for: the value launches the library is asked about are then
        #  the ones the plan makes -- planes of S - h0*w0 tokens per frame)
        nl_, vmode_ = arch.nl, str(self.opt.value_planes)
        self._p3raw_wanted = bool(vmode_ == "2" and self.opt.p3_raw and nl_ >= 2 and arch.head_ch[0] == 128 and min(self.shapes[0]) >= 2
                                  and self.shapes[0][0] * self.shapes[0][1] * head_src[0][0].ld * self._esz <= 0x7fffffff)
        fold = self._fold_plan(head_src, valid[0, :, 0])
        self.fold_proj = fold is not None
        feats = None
        if fold is None:
            feats = View(self._buf(B * S, hd))
            off = 0
            for li, ((src_view, src_hw), (h_, w_)) in enumerate(zip(head_src, self.shapes)):
                cin = arch.head_ch[li]
                Wt = self._weight(sd[f"{d}.input_proj.{li}.0.weight"].reshape(hd, cin))
                scale, shift = self._bn(f"{d}.input_proj.{li}.1")
                dst = View(feats.buf[off:], 0, hd) if off else feats
                self._gemm(src_view, Wt, hd, cin, dst, B * h_ * w_, scale=scale, shift=shift, c_rpb=h_ * w_, c_bstride=S)
                off += h_ * w_
        self.feats = feats

        # value projections of all decoder layers in ONE GEMM over the S tokens: feats is layer
        # invariant (transformer.py:700-706 passes the same `feats` to every layer)
        ndl = arch.ndl
        Wv = torch.cat([sd[f"{d}.decoder.layers.{i}.cross_attn.value_proj.weight"] for i in range(ndl)], 0)
        bv = torch.cat([sd[f"{d}.decoder.layers.{i}.cross_attn.value_proj.bias"] for i in range(ndl)], 0)
        # ... written as ndl contiguous [B*S, hd] planes: a layer's slice is dense in HBM for its deformable gather
        vmode = str(self.opt.value_planes)
        dh = hd // arch.nh
        # Round 5: level 0 (P3) of the deformable attention gathered RAW and projected after the bilinear sum (csrc/msda_raw.hip): its
        # value planes -- the longest launch of the plan, 54 % of whose output no sample touches -- are never formed.  Folded head only
        # (the composed weights are the fold's); MOY_P3_RAW=0 keeps the planes of all levels.
        self.p3raw = None
        self.value_tokens = S                  # tokens per frame in the value planes
        if fold is not None and self._p3raw_wanted:
            self.value_tokens = S - self.shapes[0][0] * self.shapes[0][1]
        if vmode == "2":
            # [layer][head][token][32]: a head's map is a dense [B*S, 32] matrix, so the two x-taps of a bilinear sample are
            # 128 contiguous bytes and the GEMM's stores are contiguous runs
            Sv = self.value_tokens
            self.value_planes = self._buf(ndl * arch.nh * B * Sv, dh)
            value = [(View(self.value_planes[i * arch.nh * B * Sv:(i * arch.nh + 1) * B * Sv]), B * Sv * dh) for i in range(ndl)]
            if fold is None:
                self._gemm(feats, self._weight(Wv), ndl * hd, hd, value[0][0], B * S, shift=self._dev(bv), planes=(dh, B * S * dh))
                self._value_launches.append((self._steps[-1][1][0]._obj, S))
                self._value_step = len(self._steps) - 1
            else:
                # value = (P . Wp^T * s + t) . Wv^T + bv = P . (Wv diag(s) Wp)^T + (Wv t + bv): one launch per level, rows (b, i) of the
                # level -> token b*S + off + i of every head plane (output row remap)
                off = 0
                for li, ((src_view, _), (h_, w_)) in enumerate(zip(head_src, self.shapes)):
                    Wc = (Wv.double() @ fold["sWp"][li]).float()
                    bc = (Wv.double() @ fold["t"][li] + bv.double()).float()
                    if li == 0 and Sv != S:
                        # level 0 is sampled raw: its composed weights [ndl * 256, 128] / biases go to the gather, layer by layer
                        if self.dtype == torch.float32:
                            # fp32 engines (round 6): the gather projects on the exact fp32 matrix instruction from fp32 row-major weights
                            self.p3raw = dict(view=src_view, wc=self._dev(Wc.contiguous()), bc=self._dev(bc))
                            continue
                        self.p3raw = dict(view=src_view, wc=self._weight(Wc), bc=self._dev(bc))
                        assert self.p3raw["wc"].shape == (ndl * hd, 128)
                        from .ops import pack_mfma_a
                        self.p3raw["wc_packed"] = pack_mfma_a(self.p3raw["wc"])     # a layer's [256, 128] slice stays at the same byte offset
                        continue
                    self._gemm(src_view, self._weight(Wc), ndl * hd, arch.head_ch[li], View(self.value_planes[off:]), B * h_ * w_,
                               shift=self._dev(bc), planes=(dh, B * Sv * dh), c_rpb=h_ * w_, c_bstride=Sv)
                    self._value_launches.append((self._steps[-1][1][0]._obj, h_ * w_))
                    if li == 0:
                        self._value_p3_step = len(self._steps) - 1
                    off += h_ * w_
        elif vmode == "1":      # A/B: one contiguous [B*S, hd] matrix per layer
            self.value_planes = self._buf(ndl * B * S, hd)
            value = [(View(self.value_planes[i * B * S:(i + 1) * B * S]), dh) for i in range(ndl)]
            self._gemm(feats, self._weight(Wv), ndl * hd, hd, value[0][0], B * S, shift=self._dev(bv), planes=(hd, B * S * hd))
        else:                   # A/B: one [B*S, ndl*hd] matrix, a layer = a 512-byte column slice of 3072-byte rows
            vall = View(self._buf(B * S, ndl * hd))
            self._gemm(feats, self._weight(Wv), ndl * hd, hd, vall, B * S, shift=self._dev(bv))
            value = [(vall.slice(i * hd, hd), dh) for i in range(ndl)]
        self.value = value

        # enc_output (Linear + LayerNorm, head.py:1036-1040) is row-wise and only the nq selected rows are ever read again
        # (head.py:1048-1113), so the pass over all S tokens produces the enc_score_head logits ONLY (C = NULL): no [B*S, 256]
        # write.  The selected rows are recomputed below by the same GEMM on gathered rows.
        Wt_enc, bias_enc = self._linear_w(d + ".enc_output.0")
        ln_enc = (self._dev(sd[d + ".enc_output.1.weight"]), self._dev(sd[d + ".enc_output.1.bias"]))
        self.scores_all = self._buf(B * S, nc, torch.float32, static=True)     # (masked tokens keep the constant written at plan build)
        wsc, bsc = self._dev(sd[d + ".enc_score_head.weight"]), self._dev(sd[d + ".enc_score_head.bias"])
        fuse_score = nc <= 8     # enc_score_head rides on the LayerNorm epilogue
        if fold is not None:
            # score pass per level over its valid rectangle, from the level's own tensor: LN((P . Wp^T * s + t) . We^T + be) . w + b with
            # composed weights; a masked token's feature row is zero (head.py:1039), so its score is the constant of LN(be): written once
            We, be = sd[d + ".enc_output.0.weight"].double(), sd[d + ".enc_output.0.bias"].double()
            const = torch.nn.functional.layer_norm(be.float()[None], (hd,), sd[d + ".enc_output.1.weight"], sd[d + ".enc_output.1.bias"], 1e-5)
            const = const @ sd[d + ".enc_score_head.weight"].t() + sd[d + ".enc_score_head.bias"]
            self.scores_all.copy_(const.to(self.dev).expand(B * S, nc))
            off = 0
            for li, ((src_view, _), (h_, w_)) in enumerate(zip(head_src, self.shapes)):
                y0, y1, x0, x1 = fold["rect"][li]
                Wc = (We @ fold["sWp"][li]).float()
                bc = (We @ fold["t"][li] + be).float()
                nv = (y1 - y0 + 1) * (x1 - x0 + 1)
                self._gemm(src_view, self._weight(Wc), hd, arch.head_ch[li], None, B * S, shift=self._dev(bc), ln=ln_enc,
                           dot=(wsc, bsc, self.scores_all), meta_scale=nv / S,
                           runs=dict(period=S, tok0=off + y0 * w_ + x0, pitch=w_, len=x1 - x0 + 1, rows=y1 - y0 + 1, a_period=h_ * w_, a_off=off))
                self.meta[-1]["name"] += f" level {li} valid-runs {nv}/{h_ * w_}"
                off += h_ * w_
        elif fuse_score:
            self._gemm(feats, Wt_enc, hd, hd, None, B * S, shift=bias_enc, a_mask=self.valid, mask_period=S, ln=ln_enc,
                       dot=(wsc, bsc, self.scores_all))
            self._score_runs(valid[0, :, 0])
        else:
            allf = View(self._buf(B * S, hd))
            self._gemm(feats, Wt_enc, hd, hd, allf, B * S, shift=bias_enc, a_mask=self.valid, mask_period=S, ln=ln_enc)
            self._add(lib.moy_rowdot, allf.ptr, allf.ld, None, B * S, hd, wsc.data_ptr(), bsc.data_ptr(), nc, 0,
                      None, None, self.scores_all.data_ptr(), code,
                      meta=dict(name=f"rowdot M{B * S} N{nc}", bytes=B * S * (hd * self._esz + nc * 4), flops=2 * B * S * hd * nc))

        self.topk_local = torch.zeros(B, nq, device=self.dev, dtype=torch.int32)
        self.topk_global = torch.zeros(B, nq, device=self.dev, dtype=torch.int32)
        self.n_masked = torch.zeros(B, device=self.dev, dtype=torch.int32)
        self._topk_step = len(self._steps)
        self._split = len(self._steps)      # [0, _split): bandwidth-bound launches over all tokens; [_split, end): the query-sized chain
        self._add(lib.moy_topk, self.scores_all.data_ptr(), B, S, nc, nq, self.valid.data_ptr(),
                  self.topk_local.data_ptr(), self.topk_global.data_ptr(), self.n_masked.data_ptr(),
                  meta=dict(name=f"topk B{B} S{S} k{nq}", bytes=B * (S * nc * 4 + S + nq * 8), flops=0))

        M = B * nq
        n_max = self.n_max
        Lq = nq + n_max                  # decoder rows per batch element: [track slots | detect queries] in temporal mode
        Md = B * Lq
        if Lq > 512:
            raise NotImplementedError("track slots + queries must not exceed 512 decoder rows (moy_mha_core)")
        t1, t2 = View(self._buf(Md, hd)), View(self._buf(Md, hd))

        def bbox_mlp(prefix, x: View, a_rows, mode, aux, aux_rows, out_t, M=M):
            W0, b0 = self._linear_w(prefix + ".layers.0")
            W1, b1 = self._linear_w(prefix + ".layers.1")
            w2, b2 = self._dev(sd[prefix + ".layers.2.weight"]), self._dev(sd[prefix + ".layers.2.bias"])
            if self.dtype != torch.float32 and self.opt.mlp_head:
                # the whole head in one launch (csrc/mlp_head.hip): hidden activations stay in LDS
                self._add(lib.moy_mlp_head, x.ptr, x.ld, a_rows.data_ptr() if a_rows is not None else None, M, W0.data_ptr(),
                          b0.data_ptr(), W1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), mode, aux.data_ptr(),
                          aux_rows.data_ptr() if aux_rows is not None else None, out_t.data_ptr(), code,
                          meta=dict(name=f"mlp_head M{M}", bytes=(M * hd + 2 * hd * hd) * 2 + M * 16, flops=4 * M * hd * hd))
                return
            self._gemm(x, W0, hd, hd, t1, M, shift=b0, act=L.ACT_RELU, a_rows=a_rows)
            self._gemm(t1, W1, hd, hd, t2, M, shift=b1, act=L.ACT_RELU)
            self._add(lib.moy_rowdot, t2.ptr, t2.ld, None, M, hd, w2.data_ptr(), b2.data_ptr(), 4, mode,
                      aux.data_ptr(), aux_rows.data_ptr() if aux_rows is not None else None, out_t.data_ptr(), code,
                      meta=dict(name=f"rowdot M{M} N4", bytes=M * (hd * self._esz + 32), flops=2 * M * hd * 4))

        # embed[0] = decoder input (never overwritten by the step), embed[1], embed[2] = layer outputs (ping-pong)
        embed = [View(self._buf(Md, hd)) for _ in range(3)]
        qpos = View(self._buf(Md, hd))
        refs = [self._buf(Md, 4, torch.float32) for _ in range(2)]
        # enc_output of the selected tokens = the decoder's content queries (head.py:1104-1113)
        sel = embed[0] if not n_max else View(self._buf(M, hd))
        if fold is None:
            self._gemm(feats, Wt_enc, hd, hd, sel, M, shift=bias_enc, a_rows=self.topk_global, a_mask=self.valid, mask_period=S,
                       ln=ln_enc)
        else:
            # the projected features of the nq selected tokens only: token -> (level, row in the level's tensor); one gathered product per
            # level (fp32, BN scale applied), the row's own level kept + its BN shift, rounded once; masked tokens -> zero rows
            nl_ = len(self.shapes)
            self.sel_rows_lvl = torch.zeros(nl_, M, device=self.dev, dtype=torch.int32)
            self.sel_level = torch.zeros(M, device=self.dev, dtype=torch.int32)
            hw_c = (C.c_int32 * nl_)(*[h_ * w_ for h_, w_ in self.shapes])
            self._keep.append(hw_c)
            self._add(lib.moy_level_rows, self.topk_local.data_ptr(), B, nq, nl_, hw_c, self.sel_rows_lvl.data_ptr(), self.sel_level.data_ptr(),
                      meta=dict(name=f"level_rows M{M}", bytes=M * (4 + 4 * nl_ + 4), flops=0))
            G = self._buf(nl_ * M, hd, torch.float32)
            shifts = self._dev(torch.stack([t_.float() for t_ in fold["t"]]))
            for li, (src_view, _) in enumerate(head_src):
                cin = arch.head_ch[li]
                Wt = self._weight(sd[f"{d}.input_proj.{li}.0.weight"].reshape(hd, cin))
                self._gemm(src_view, Wt, hd, cin, View(G[li * M:(li + 1) * M]), M, scale=self._dev(fold["s"][li].float()),
                           a_rows=self.sel_rows_lvl[li], out_f32=True)
            fsel = View(self._buf(M, hd))
            self._add(lib.moy_level_select, G.data_ptr(), M * hd, hd, self.sel_level.data_ptr(), shifts.data_ptr(), self.topk_local.data_ptr(),
                      self.valid.data_ptr(), M, hd, fsel.ptr, fsel.ld, code,
                      meta=dict(name=f"level_select M{M}", bytes=M * hd * (4 + self._esz) + M * 8, flops=0))
            self.feats_selected = fsel
            self._gemm(fsel, Wt_enc, hd, hd, sel, M, shift=bias_enc, ln=ln_enc)
        self.features = sel              # [B*nq, 256]: rows of enc_output at the selected tokens, in query order
        self.refer_logit = self._buf(M, 4, torch.float32)
        bbox_mlp(d + ".enc_bbox_head", sel, None, 2, self.anchors, self.topk_local, self.refer_logit)

        if not n_max:
            self._add(lib.moy_pos2posemb, self.refer_logit.data_ptr(), M, qpos.ptr, qpos.ld, code,
                      meta=dict(name=f"pos2posemb M{M}", bytes=M * (16 + hd * self._esz), flops=0))
            self._add(lib.moy_sigmoid_f32, self.refer_logit.data_ptr(), M * 4, refs[0].data_ptr(),
                      meta=dict(name=f"sigmoid M{M}", bytes=M * 32, flops=0))
            self.refer_all = self.refer_logit
        else:
            # per-sequence query memory (static shapes: the whole temporal step stays graph-capturable)
            i32, i64 = torch.int32, torch.int64
            self.trk = dict(embed=self._buf(B * n_max, hd, static=True), qpos=self._buf(B * n_max, hd, static=True),
                            ref=self._buf(B * n_max, 4, torch.float32, static=True),
                            id=torch.full((B, n_max), -1, device=self.dev, dtype=i64),
                            dis=torch.zeros(B, n_max, device=self.dev, dtype=i32),
                            n=torch.zeros(B, device=self.dev, dtype=i32),
                            max_obj_id=torch.zeros(B, device=self.dev, dtype=i64))
            det_embed, det_qpos = sel, View(self._buf(M, hd))
            self._add(lib.moy_pos2posemb, self.refer_logit.data_ptr(), M, det_qpos.ptr, det_qpos.ld, code,
                      meta=dict(name=f"pos2posemb M{M}", bytes=M * (16 + hd * self._esz), flops=0))
            self.refer_all = self._buf(Md, 4, torch.float32)
            t = self.trk
            self._add(lib.moy_temporal_assemble, t["embed"].data_ptr(), t["qpos"].data_ptr(), t["ref"].data_ptr(), t["n"].data_ptr(),
                      det_embed.ptr, det_embed.ld, det_qpos.ptr, det_qpos.ld, self.refer_logit.data_ptr(), B, n_max, nq,
                      embed[0].ptr, embed[0].ld, qpos.ptr, qpos.ld, self.refer_all.data_ptr(), refs[0].data_ptr(), code,
                      meta=dict(name=f"temporal_assemble B{B} L{nq + n_max}", bytes=Md * (4 * hd * self._esz + 32), flops=0))
        self.query_pos = qpos
        # Round 6: the deformable gather walks a frame's queries in the Morton order of their reference points' P3 cells (one sort per
        # frame, of the decoder's initial boxes: the six refinements move them little, transformer.py:676-728), so that the queries one
        # block gathers for are neighbours; rows and bits of every output are unchanged (MOY_Q_ORDER=0: top-k order, as rounds 1-5)
        self.qperm = None
        if self.p3raw is not None and self.opt.query_order and Lq <= 1024:
            self.qperm = torch.zeros(B, Lq, device=self.dev, dtype=torch.int32)
            self._add(lib.moy_query_order, refs[0].data_ptr(), B, Lq, self.shapes[0][0], self.shapes[0][1], self.qperm.data_ptr(),
                      meta=dict(name=f"query_order B{B} L{Lq}", bytes=Md * 20, flops=0))

        M = Md                           # from here on: decoder rows
        qkv = View(self._buf(M, 3 * hd))
        attn = View(self._buf(M, hd))
        e1, e2 = View(self._buf(M, hd)), View(self._buf(M, hd))
        offaw = self._buf(M, 8 * nl * 4 * 3, torch.float32)
        samp = View(self._buf(M, hd))
        ffn = View(self._buf(M, arch.d_ffn))
        self.debug_views = dict(qkv=qkv.tensor(), attn=attn.tensor(), e1=e1.tensor(), offaw=offaw, samp=samp.tensor())   # tools/probes
        shapes_c = (C.c_int32 * (2 * nl))(*[v for hw_ in self.shapes for v in hw_])
        self._keep.append(shapes_c)
        self.layer_out = []
        cur, nxt = 0, 1
        ecur, enxt = 0, 1
        # x + query_pos of the next layer, written by the fused tail (16-bit engines), at the sizes where the plain products take the
        # weight-stationary kernel (MOY_QKV_SPLIT=0 switches it off, 2 forces it at any size).  Round 3 measured -0.2 ms on the sum of a
        # pass's kernels and +-0 on the two-stream step and left it off; round 5 (two interleaved pairs on one device): 33.45 / 33.33 ->
        # 33.24 / 33.28 ms per step, on by default
        use_xp = (self.dtype != torch.float32 and arch.d_ffn % 256 == 0 and self.opt.dec_tail
                  and self.opt.qkv_split != 0 and (M >= 65536 or self.opt.qkv_split == 2))
        # round 5, OFF by default (MOY_QKV_FUSE=1): the fused tail of layer i can also PROJECT q | k | v of layer i + 1 while its rows are on
        # chip, instead of the two plain products over `xp`.  Bit-identical (tests), 10 launches fewer per pass -- and measured neutral: the three
        # extra products cost the tail what the two launches cost (sum of a pass's kernels 17.795 against 17.771 ms, step 64.40 / 64.51 against
        # 64.25 / 64.76 ms, two interleaved pairs): 1.4 MB of weights per 128-row tile is the tail's price per product, the plain products pay it once
        fuse_qkv = use_xp and self.opt.qkv_fuse
        # round 6: at SMALL M (the small-batch leg: a few frames per step) a launch is its own latency, not its work: the tail projects the
        # next layer's q | k | v while its rows are on chip -- one launch less per layer; bit-identical (tests)
        if (not use_xp and self.opt.qkv_fuse_small and M < 65536 and self.dtype != torch.float32 and arch.d_ffn % 256 == 0 and self.opt.dec_tail):
            fuse_qkv = True
        qkv_ready = False
        xp = View(self._buf(M, hd)) if use_xp and not fuse_qkv else None
        from .ops import pack_mfma_a
        packed_w = self.opt.w_packed       # the row-wise decoder kernels' weights in MFMA-fragment order
        xp_ready = False
        for i in range(ndl):
            q = f"{d}.decoder.layers.{i}"
            # q | k | v in one launch: q = k = x + pos for the first 2*hd columns, v = x for the rest (transformer.py:637-640)
            x = embed[ecur]
            if i > 0 and qkv_ready:
                pass                                   # the previous layer's tail wrote q | k | v of this layer
            elif i > 0 and xp_ready:
                # the previous layer's tail left x + query_pos in `xp`: q | k and v as two PLAIN products (at bench scale they take
                # the weight-stationary kernel, which has no second A operand)
                Wi, bi = sd[q + ".self_attn.in_proj_weight"], sd[q + ".self_attn.in_proj_bias"]
                Wqk, bqk = self._linear_w_raw(Wi[:2 * hd], bi[:2 * hd])
                Wv_, bv_ = self._linear_w_raw(Wi[2 * hd:], bi[2 * hd:])
                self._gemm(xp, Wqk, 2 * hd, hd, qkv.slice(0, 2 * hd), M, shift=bqk)
                self._gemm(x, Wv_, hd, hd, qkv.slice(2 * hd, hd), M, shift=bv_)
            else:
                Wqkv, bqkv = self._linear_w_raw(sd[q + ".self_attn.in_proj_weight"], sd[q + ".self_attn.in_proj_bias"])
                self._gemm(x, Wqkv, 3 * hd, hd, qkv, M, shift=bqkv, A2=qpos, a2_cols=2 * hd)
            if n_max:   # keys: live track slots + this frame's detect queries
                self._add(lib.moy_mha_core_masked, qkv.ptr, qkv.ld, B, Lq, arch.nh, hd, self.trk["n"].data_ptr(), n_max, attn.ptr,
                          attn.ld, code, meta=dict(name=f"mha_core B{B} L{Lq}", bytes=4 * M * hd * self._esz, flops=4 * B * Lq * Lq * hd))
            else:
                self._add(lib.moy_mha_core, qkv.ptr, qkv.ld, B, nq, arch.nh, hd, attn.ptr, attn.ld, code,
                          meta=dict(name=f"mha_core B{B} L{nq}", bytes=4 * M * hd * self._esz, flops=4 * B * nq * nq * hd))
            Wo, bo = self._linear_w(q + ".self_attn.out_proj")
            Woa = torch.cat([sd[q + ".cross_attn.sampling_offsets.weight"], sd[q + ".cross_attn.attention_weights.weight"]], 0)
            boa = torch.cat([sd[q + ".cross_attn.sampling_offsets.bias"], sd[q + ".cross_attn.attention_weights.bias"]], 0)
            n_oa = Woa.shape[0]
            if self.dtype != torch.float32 and n_oa % 32 == 0 and n_oa <= 512 and self.opt.dec_mid:
                # out_proj + norm1 and the offsets | weights linear of (e1 + query_pos) as ONE launch (csrc/dec_mid.hip)
                Wpad = torch.zeros(max(256, n_oa), hd)
                Wpad[:n_oa] = Woa
                Woa_d, boa_d = self._weight(Wpad), self._dev(boa)
                ln1 = self._ln(q + ".norm1")
                t = L.DecoderMidArgs()
                if packed_w:
                    Wo, Woa_d = pack_mfma_a(Wo), pack_mfma_a(Woa_d)
                    t.w_packed = 1
                t.attn, t.ld_attn, t.x, t.ld_x, t.qpos, t.ld_qpos, t.M = attn.ptr, attn.ld, x.ptr, x.ld, qpos.ptr, qpos.ld, M
                t.Wo, t.bo, t.ln_g, t.ln_b = Wo.data_ptr(), bo.data_ptr(), ln1[0].data_ptr(), ln1[1].data_ptr()
                t.Woa, t.boa, t.n_oa = Woa_d.data_ptr(), boa_d.data_ptr(), n_oa
                t.e1, t.ld_e1, t.offaw, t.ld_oa, t.dtype = e1.ptr, e1.ld, offaw.data_ptr(), offaw.shape[1], code
                self._keep += [t, Wo, Woa_d]
                self._add(lib.moy_decoder_mid, C.byref(t),
                          meta=dict(name=f"decoder_mid M{M}", bytes=4 * M * hd * 2 + M * n_oa * 4 + (hd + n_oa) * hd * 2,
                                    flops=2 * M * hd * (hd + n_oa)))
            else:
                self._gemm(attn, Wo, hd, hd, e1, M, shift=bo, R=x, ln=self._ln(q + ".norm1"))
                Woa_d, boa_d = self._linear_w_raw(Woa, boa)
                self._gemm(e1, Woa_d, n_oa, hd, View(offaw), M, shift=boa_d, A2=qpos, out_f32=True)
            vslice, vhs = value[i]
            # gather: SURVEY §8(d) charges min(value slice of the layer, taps x 64 B); the taps of a launch fall on the same cells
            # (one cell = one token of one head = 32 channels) again and again, so what has to MOVE is the expected number of
            # DISTINCT cells under uniform sampling, cells x (1 - exp(-taps / cells)) -- 1.31 GB per launch at 288 frames against
            # 1.20 GB measured by PMC (round 2 charged 2.15 GB and the launch table showed a fraction > 1)
            import math
            if self.p3raw is not None:
                # levels 1..: distinct 64-byte head cells as before; level 0: distinct 256-byte PIXELS of the raw map (every head's taps
                # read whole pixels) + the layer's composed weights once
                hw0 = self.shapes[0][0] * self.shapes[0][1]
                cells1, taps1 = B * self.value_tokens * arch.nh, M * arch.nh * (nl - 1) * arch.ndp * 4
                cells0, taps0 = B * hw0, M * arch.nh * arch.ndp * 4
                touched = (int(cells1 * (1.0 - math.exp(-taps1 / cells1))) * dh + int(cells0 * (1.0 - math.exp(-taps0 / cells0))) * 128) * self._esz
                t = L.MsdaRawArgs()
                pv = self.p3raw["view"]
                t.x0, t.ld0 = pv.ptr, pv.ld
                pk = packed_w and "wc_packed" in self.p3raw
                t.wc = self.p3raw["wc_packed" if pk else "wc"].data_ptr() + i * hd * 128 * self._esz
                t.wc_packed = int(pk)
                t.bc = self.p3raw["bc"].data_ptr() + i * hd * 4
                t.planes, t.head_stride, t.S1 = vslice.ptr, vhs, self.value_tokens
                t.B, t.Lq, t.L, t.shapes_hw = B, Lq, nl, C.cast(shapes_c, C.c_void_p)
                t.offaw, t.ld_oa, t.ref, t.out, t.ldo, t.dtype = offaw.data_ptr(), offaw.shape[1], refs[cur].data_ptr(), samp.ptr, samp.ld, self.code_gemm
                t.perm = self.qperm.data_ptr() if self.qperm is not None else None
                self._keep.append(t)
                self._add(lib.moy_msda_raw0, C.byref(t),
                          meta=dict(name=f"msda_raw0 M{M}", bytes=touched + hd * 128 * self._esz + M * (offaw.shape[1] * 4 + 16 + hd * self._esz),
                                    flops=2 * M * arch.nh * nl * arch.ndp * 4 * dh + 2 * M * hd * 128 + 2 * M * arch.nh * arch.ndp * 4 * (128 - dh)))
            else:
                cells = B * S * arch.nh
                ntaps = M * arch.nh * nl * arch.ndp * 4
                touched = int(cells * (1.0 - math.exp(-ntaps / cells))) * (hd // arch.nh) * self._esz
                self._add(lib.moy_msda_fused, vslice.ptr, vslice.ld, vhs, B, S, shapes_c, nl, offaw.data_ptr(), offaw.shape[1],
                          refs[cur].data_ptr(), Lq, samp.ptr, samp.ld, code,
                          meta=dict(name=f"msda_fused M{M}", bytes=touched + M * (offaw.shape[1] * 4 + 16 + hd * self._esz),
                                    flops=2 * M * arch.nh * nl * arch.ndp * 4 * (hd // arch.nh)))
            Wp, bp = self._linear_w(q + ".cross_attn.output_proj")
            W1, b1 = self._linear_w(q + ".linear1")
            W2, b2 = self._linear_w(q + ".linear2")
            if self.dtype != torch.float32 and arch.d_ffn % 256 == 0 and self.opt.dec_tail:
                # output_proj + norm2, FFN + norm3 and the box refinement as one launch (csrc/dec_tail.hip)
                ln2, ln3 = self._ln(q + ".norm2"), self._ln(q + ".norm3")
                hp = f"{d}.dec_bbox_head.{i}"
                B0, c0 = self._linear_w(hp + ".layers.0")
                B1, c1 = self._linear_w(hp + ".layers.1")
                w2h, c2 = self._dev(sd[hp + ".layers.2.weight"]), self._dev(sd[hp + ".layers.2.bias"])
                t = L.DecoderTailArgs()
                if packed_w:     # MFMA-fragment order (include/moyolo.h): a wave's weight request is 2 KB contiguous, not 32 rows x 64 bytes
                    Wp, W1, W2, B0, B1 = (pack_mfma_a(w) for w in (Wp, W1, W2, B0, B1))
                    t.w_packed = 1
                t.samp, t.ld_samp, t.e1, t.ld_e1, t.M = samp.ptr, samp.ld, e1.ptr, e1.ld, M
                t.Wp, t.bp, t.ln2_g, t.ln2_b = Wp.data_ptr(), bp.data_ptr(), ln2[0].data_ptr(), ln2[1].data_ptr()
                t.W1, t.b1, t.W2, t.b2, t.d_ffn = W1.data_ptr(), b1.data_ptr(), W2.data_ptr(), b2.data_ptr(), arch.d_ffn
                t.ln3_g, t.ln3_b, t.out, t.ld_out = ln3[0].data_ptr(), ln3[1].data_ptr(), embed[enxt].ptr, embed[enxt].ld
                t.B0, t.c0, t.B1, t.c1, t.w2, t.c2 = (B0.data_ptr(), c0.data_ptr(), B1.data_ptr(), c1.data_ptr(), w2h.data_ptr(),
                                                      c2.data_ptr())
                t.ref_in, t.ref_out, t.dtype = refs[cur].data_ptr(), refs[nxt].data_ptr(), code
                wq_next = None
                if fuse_qkv and i + 1 < ndl:
                    qn = f"{d}.decoder.layers.{i + 1}"
                    wq_next, bq_next = self._linear_w_raw(sd[qn + ".self_attn.in_proj_weight"], sd[qn + ".self_attn.in_proj_bias"])
                    if packed_w:
                        wq_next = pack_mfma_a(wq_next)
                    t.Wqkv, t.bqkv, t.qkv, t.ld_qkv = wq_next.data_ptr(), bq_next.data_ptr(), qkv.ptr, qkv.ld
                    t.qpos, t.ld_qpos = qpos.ptr, qpos.ld
                    self._keep += [wq_next, bq_next]
                    qkv_ready = True
                elif use_xp and i + 1 < ndl:
                    t.qpos, t.ld_qpos, t.out_xp, t.ld_xp = qpos.ptr, qpos.ld, xp.ptr, xp.ld
                    xp_ready = True
                self._keep += [t, Wp, W1, W2, B0, B1]
                wb = (3 * hd * hd + 2 * hd * arch.d_ffn + (3 * hd * hd if wq_next is not None else 0)) * 2
                self._add(lib.moy_decoder_tail, C.byref(t),
                          meta=dict(name=f"decoder_tail{'+qkv' if wq_next is not None else ''} M{M}",
                                    bytes=(3 + (4 if wq_next is not None else 0)) * M * hd * 2 + wb + M * 32,
                                    flops=2 * M * hd * (3 * hd + 2 * arch.d_ffn + (3 * hd if wq_next is not None else 0))))
            else:
                self._gemm(samp, Wp, hd, hd, e2, M, shift=bp, R=e1, ln=self._ln(q + ".norm2"))
                self._gemm(e2, W1, arch.d_ffn, hd, ffn, M, shift=b1, act=L.ACT_RELU)
                self._gemm(ffn, W2, hd, arch.d_ffn, embed[enxt], M, shift=b2, R=e2, ln=self._ln(q + ".norm3"))
                bbox_mlp(f"{d}.dec_bbox_head.{i}", embed[enxt], None, 1, refs[cur], None, refs[nxt], M=M)
            self.layer_out.append((embed[enxt], refs[nxt]))
            cur, nxt = nxt, cur
            ecur, enxt = enxt, (2 if enxt == 1 else 1)
        self.hs = embed[ecur]
        self.boxes = refs[cur]
        self.logits = self._buf(M, nc, torch.float32)
        wd = self._dev(sd[f"{d}.dec_score_head.{ndl - 1}.weight"])
        bd = self._dev(sd[f"{d}.dec_score_head.{ndl - 1}.bias"])
        self._add(lib.moy_rowdot, self.hs.ptr, self.hs.ld, None, M, hd, wd.data_ptr(), bd.data_ptr(), nc, 0, None, None,
                  self.logits.data_ptr(), code,
                  meta=dict(name=f"rowdot M{M} N{nc}", bytes=M * (hd * self._esz + nc * 4), flops=2 * M * hd * nc))

        self.y = torch.zeros(B, Lq, 4 + nc, device=self.dev)
        self.scores = torch.zeros(B, Lq, device=self.dev)
        self.obj_idxes = torch.zeros(B, Lq, device=self.dev, dtype=torch.int64)
        self._alloc_result_block(Lq, with_ids=True)
        if not n_max:
            self._add(lib.moy_assign_post, self.logits.data_ptr(), self.boxes.data_ptr(), B, nq, nc,
                      C.c_float(self.score_thresh), C.c_float(self.conf), C.c_float(self.img_wh[0]), C.c_float(self.img_wh[1]),
                      self.y.data_ptr(), self.scores.data_ptr(), self.obj_idxes.data_ptr(), self.rows.data_ptr(),
                      self.track_id.data_ptr(), self.n_rows.data_ptr(), self.n_ids.data_ptr(),
                      meta=dict(name=f"assign_post B{B} nq{nq}", bytes=B * nq * ((nc + 4) * 4 + (4 + nc) * 4 + 4 + 8 + 24 + 8), flops=0))
        else:
            self._build_temporal_update(qpos)

        if self.side_state:
            i32, i64, f32 = torch.int32, torch.int64, torch.float32
            z = lambda *shape, dt=f32: torch.zeros(*shape, device=self.dev, dtype=dt)
            self.copy_rows, self.copy_ids, self.n_copy = z(B, nq, dt=i32), z(B, nq, dt=i64), z(B, dt=i32)
            self.fsqm = dict(mem=z(300, 256), conf=z(300), ids=z(300, dt=i64), boxes=z(300, 4), low=z(300, dt=i32),
                             pool=z(1 << 16, dt=i32), pool_hc=z(3, dt=i32))
            self.reset_sequence()
            f = self.fsqm
            self._add(lib.moy_track_state_update, self.scores.data_ptr(), self.boxes.data_ptr(), self.obj_idxes.data_ptr(),
                      self.hs.ptr, self.hs.ld, B, nq, self.copy_rows.data_ptr(), self.copy_ids.data_ptr(), self.n_copy.data_ptr(),
                      f["mem"].data_ptr(), f["conf"].data_ptr(), f["ids"].data_ptr(), f["boxes"].data_ptr(), f["low"].data_ptr(),
                      f["pool"].data_ptr(), f["pool"].numel(), f["pool_hc"].data_ptr(), code,
                      meta=dict(name=f"track_state_update B{B} Q{nq}", bytes=B * nq * (28 + hd * self._esz) + 300 * (hd * 4 + 40), flops=0))

    def _build_temporal_update(self, qpos: View):
        """After the decoder of a frame (temporal mode): ID lifecycle + compaction (`moy_temporal_assign`), the learned
        query update of the surviving / newborn tracks (`QueryInteractionModule._update_track_embedding`,
        MOTR/models/qim.py:251-301, update_query_pos=False) on the compacted slots, commit into the query memory."""
        arch, B, sd, lib, code = self.arch, self.B, self.sd, self.lib, self.code
        hd, nq, nc, n_max = arch.hd, arch.nq, arch.nc, self.n_max
        Lq, Mq = nq + n_max, B * n_max
        i32, i64 = torch.int32, torch.int64
        t = self.trk
        self.dis_out = torch.zeros(B, Lq, device=self.dev, dtype=i32)
        self.sel_rows = torch.zeros(B, n_max, device=self.dev, dtype=i32)
        self.n_new = torch.zeros(B, device=self.dev, dtype=i32)
        self.n_overflow = torch.zeros(B, device=self.dev, dtype=i32)
        self._add(lib.moy_temporal_assign, self.logits.data_ptr(), self.boxes.data_ptr(), B, n_max, nq, nc, t["id"].data_ptr(),
                  t["dis"].data_ptr(), t["n"].data_ptr(), t["max_obj_id"].data_ptr(), C.c_float(self.score_thresh),
                  C.c_float(self.filter_score_thresh), self.miss_tolerance, C.c_float(self.conf), C.c_float(self.img_wh[0]),
                  C.c_float(self.img_wh[1]), self.y.data_ptr(), self.scores.data_ptr(), self.obj_idxes.data_ptr(),
                  self.dis_out.data_ptr(), self.sel_rows.data_ptr(), self.n_new.data_ptr(), self.n_overflow.data_ptr(),
                  self.rows.data_ptr(), self.track_id.data_ptr(), self.n_rows.data_ptr(), self.n_ids.data_ptr(),
                  meta=dict(name=f"temporal_assign B{B} L{Lq}", bytes=B * Lq * (4 * nc + 16 + 24 + 12) + B * n_max * 16, flops=0))
        # the selected rows' decoder output (content embedding of the track from now on), the position embedding and the
        # reference box they were decoded with
        sel = self.sel_rows.data_ptr()
        class_content = self.track_content == "class_embed"
        if class_content and nc != 1:
            raise NotImplementedError("track_content='class_embed' needs the class of every carried row on the device: built for nc == 1 "
                                      "(every configuration of this path); the default 'decoder_output' has no such limit")
        # the selected rows' decoder output: the track's content embedding from now on (default) and, in both modes, what the learned
        # query update consumes
        out_embed = View(self._buf(Mq, hd)) if class_content else View(t["embed"])
        qp_prev = View(self._buf(Mq, hd))
        ref_in = self._buf(Mq, 4, torch.float32)
        gm = dict(name=f"gather_rows M{Mq} C{hd}", bytes=Mq * (2 * hd * self._esz + 4), flops=0)
        self._add(lib.moy_gather_rows, self.hs.ptr, self.hs.ld, sel, Mq, hd, out_embed.ptr, out_embed.ld, code, meta=gm)
        self._add(lib.moy_gather_rows, qpos.ptr, qpos.ld, sel, Mq, hd, qp_prev.ptr, qp_prev.ld, code, meta=gm)
        self._add(lib.moy_gather_rows, self.refer_all.data_ptr(), 4, sel, Mq, 4, ref_in.data_ptr(), 4, L.F32,
                  meta=dict(name=f"gather_rows M{Mq} C4", bytes=Mq * 36, flops=0))
        pos = View(self._buf(Mq, hd))
        self._add(lib.moy_pos2posemb, ref_in.data_ptr(), Mq, pos.ptr, pos.ld, code,
                  meta=dict(name=f"pos2posemb M{Mq}", bytes=Mq * (16 + hd * self._esz), flops=0))
        q = f"model.{len(arch.layers)}.track_embed"
        Wqk, bqk = self._linear_w_raw(sd[q + ".self_attn.in_proj_weight"][:2 * hd], sd[q + ".self_attn.in_proj_bias"][:2 * hd])
        Wvv, bvv = self._linear_w_raw(sd[q + ".self_attn.in_proj_weight"][2 * hd:], sd[q + ".self_attn.in_proj_bias"][2 * hd:])
        qkv = View(self._buf(Mq, 3 * hd))
        self._gemm(out_embed, Wqk, 2 * hd, hd, qkv.slice(0, 2 * hd), Mq, shift=bqk, A2=pos)
        self._gemm(out_embed, Wvv, hd, hd, qkv.slice(2 * hd, hd), Mq, shift=bvv)
        attn = View(self._buf(Mq, hd))
        self._add(lib.moy_mha_core_masked, qkv.ptr, qkv.ld, B, n_max, arch.nh, hd, self.n_new.data_ptr(), n_max, attn.ptr, attn.ld,
                  code, meta=dict(name=f"mha_core B{B} L{n_max}", bytes=4 * Mq * hd * self._esz, flops=4 * B * n_max * n_max * hd))
        tgt1, tgt2 = View(self._buf(Mq, hd)), View(self._buf(Mq, hd))
        Wo, bo = self._linear_w(q + ".self_attn.out_proj")
        self._gemm(attn, Wo, hd, hd, tgt1, Mq, shift=bo, R=out_embed, ln=self._ln(q + ".norm1"))
        W1, b1 = self._linear_w(q + ".linear1")
        W2, b2 = self._linear_w(q + ".linear2")
        dff = sd[q + ".linear1.weight"].shape[0]
        f1 = View(self._buf(Mq, dff))
        self._gemm(tgt1, W1, dff, hd, f1, Mq, shift=b1, act=L.ACT_RELU)
        self._gemm(f1, W2, hd, dff, tgt2, Mq, shift=b2, R=tgt1, ln=self._ln(q + ".norm2"))
        Wf1, bf1 = self._linear_w(q + ".linear_feat1")
        Wf2, bf2 = self._linear_w(q + ".linear_feat2")
        g1 = View(self._buf(Mq, dff))
        self._gemm(tgt2, Wf1, dff, hd, g1, Mq, shift=bf1, act=L.ACT_RELU)
        self._gemm(g1, Wf2, hd, dff, View(t["qpos"]), Mq, shift=bf2, R=qp_prev, ln=self._ln(q + ".norm_feat"))
        if class_content:
            # head.py:888-900: the carried queries' content = the class embedding of the track's class (nc == 1: row 0 for every slot;
            # dead slots are zeroed by moy_temporal_assemble whatever the memory holds)
            d_ = f"model.{len(arch.layers)}.decoder"
            table = self._dev(sd[d_ + ".denoising_class_embed.weight"].reshape(nc, hd), self.dtype)
            cls0 = torch.zeros(Mq, device=self.dev, dtype=i32)
            self._keep.append(cls0)
            self._add(lib.moy_gather_rows, table.data_ptr(), hd, cls0.data_ptr(), Mq, hd, t["embed"].data_ptr(), hd, code,
                      meta=dict(name=f"gather_rows M{Mq} C{hd} (class embedding)", bytes=Mq * hd * self._esz + hd * self._esz, flops=0))
        self._add(lib.moy_temporal_commit, sel, self.n_new.data_ptr(), self.obj_idxes.data_ptr(), self.dis_out.data_ptr(),
                  self.boxes.data_ptr(), B, n_max, t["id"].data_ptr(), t["dis"].data_ptr(), t["ref"].data_ptr(), t["n"].data_ptr(),
                  meta=dict(name=f"temporal_commit B{B} N{n_max}", bytes=Mq * (4 + 8 + 4 + 16 + 28), flops=0))

    def _build_detect_head(self, head_src):
        """Detect head of config C1 (nn/modules/head.py:27-78) + NMS / scale_boxes
        (utils/ops.py:148-283, 99-129; models/yolo/detect/predict.py:12-30)."""
        arch, B, sd, lib, code = self.arch, self.B, self.sd, self.lib, self.code
        h = f"model.{len(arch.layers)}"
        nc = arch.nc
        c2, c3 = max(16, arch.head_ch[0] // 4, 64), max(arch.head_ch[0], min(nc, 100))
        A = sum(hh * ww for _, (hh, ww) in head_src)
        self.A = A
        self.y = torch.zeros(B, 4 + nc, A, device=self.dev)
        a_off = 0
        for li, (x, (hh, ww)) in enumerate(head_src):
            M = B * hh * ww
            outs = {}
            for name, cm, cout in (("cv2", c2, 64), ("cv3", c3, nc)):
                t1, t2 = View(self._buf(M, cm)), View(self._buf(M, cm))
                self._conv(f"{h}.{name}.{li}.0", x, (hh, ww), arch.head_ch[li], cm, 3, 1, t1)
                self._conv(f"{h}.{name}.{li}.1", t1, (hh, ww), cm, cm, 3, 1, t2)
                Wt = self._weight(sd[f"{h}.{name}.{li}.2.weight"].reshape(cout, cm))
                o = View(self._buf(M, cout))
                self._gemm(t2, Wt, cout, cm, o, M, shift=self._dev(sd[f"{h}.{name}.{li}.2.bias"]))
                outs[name] = o
            stride = float(self.H // hh)
            self._add(lib.moy_detect_decode, outs["cv2"].ptr, outs["cv2"].ld, outs["cv3"].ptr, outs["cv3"].ld, B, hh, ww, nc,
                      C.c_float(stride), a_off, A, self.y.data_ptr(), code,
                      meta=dict(name=f"detect_decode M{M}", bytes=M * ((64 + nc) * self._esz + (4 + nc) * 4), flops=0))
            a_off += hh * ww
        self._alloc_result_block(self.max_det, with_ids=False)
        gain, padx, pady, cw, ch_ = 1.0, 0.0, 0.0, 0.0, 0.0
        if self.orig_hw is not None:                     # ops.scale_boxes, utils/ops.py:116-128
            oh, ow = self.orig_hw
            gain = min(self.H / oh, self.W / ow)
            padx, pady = round((self.W - ow * gain) / 2 - 0.1), round((self.H - oh * gain) / 2 - 0.1)
            cw, ch_ = float(ow), float(oh)
        self._add(lib.moy_nms, self.y.data_ptr(), B, nc, A, C.c_float(self.conf), C.c_float(self.iou), self.max_det,
                  C.c_float(7680.0), C.c_float(gain), C.c_float(padx), C.c_float(pady), C.c_float(cw), C.c_float(ch_),
                  self.rows.data_ptr(), self.n_rows.data_ptr(),
                  meta=dict(name=f"nms B{B} A{A}", bytes=B * (A * (4 + nc) * 4 + self.max_det * 24), flops=0))

    def _alloc_result_block(self, rows_per_frame: int, with_ids: bool):
        """Everything a predictor reads back per chunk -- rows [B, R, 6] f32 | track_id [B, R] i64 | n_rows [B] i32 | n_ids [B] i32
        (predict.py:43-94's TrackResults inputs) -- lives in ONE contiguous device block, so that the caller's device-to-host
        transfer is a single copy (`result_block`; `unpack_result_block` reads a host copy of it).  The reference pays one
        blocking `.cpu()` per tensor and frame (predict.py:27-76); round 3's predictor paid four per chunk."""
        B, R = self.B, rows_per_frame
        o_rows = 0
        o_tid = o_rows + B * R * 6 * 4
        o_nr = o_tid + (B * R * 8 if with_ids else 0)
        o_ni = o_nr + (B * 4 + 7) // 8 * 8
        end = o_ni + ((B * 4 + 7) // 8 * 8 if with_ids else 0)
        blk = torch.zeros(end, device=self.dev, dtype=torch.uint8)
        self.result_block = blk
        self._result_layout = dict(B=B, R=R, with_ids=with_ids, o_tid=o_tid, o_nr=o_nr, o_ni=o_ni, nbytes=end)
        self.rows = blk[o_rows:o_tid].view(torch.float32).view(B, R, 6)
        self.n_rows = blk[o_nr:o_nr + B * 4].view(torch.int32)
        if with_ids:
            self.track_id = blk[o_tid:o_nr].view(torch.int64).view(B, R)
            self.n_ids = blk[o_ni:o_ni + B * 4].view(torch.int32)

    def unpack_result_block(self, host_block):
        """Views (numpy) of a HOST copy of `result_block`: rows, track_id (or None), n_rows, n_ids (or None)."""
        import numpy as np
        lay = self._result_layout
        a = host_block.numpy() if isinstance(host_block, torch.Tensor) else host_block
        B, R = lay["B"], lay["R"]
        rows = a[:lay["o_tid"]].view(np.float32).reshape(B, R, 6)
        n_rows = a[lay["o_nr"]:lay["o_nr"] + 4 * B].view(np.int32)
        if not lay["with_ids"]:
            return rows, None, n_rows, None
        tid = a[lay["o_tid"]:lay["o_nr"]].view(np.int64).reshape(B, R)
        n_ids = a[lay["o_ni"]:lay["o_ni"] + 4 * B].view(np.int32)
        return rows, tid, n_rows, n_ids

    def _post_candidate(self, Ls, consumers, home):
        """Round 4: a down-sampling Conv (64 -> 128, 3x3, stride 2: yolo_track.yaml:19) whose ONLY consumer is the generic C2f that
        follows (i.e. its cv1, block.py:225-235) and whose output lives in no concat buffer: planned as one launch with that cv1
        (`moy_gemm_args.post_*`); MOY_POST_1X1=0 keeps the two launches."""
        if self.dtype == torch.float32 or not self.opt.post_1x1:
            return False
        if not (Ls.k == 3 and Ls.s == 2 and Ls.c1 == 64 and Ls.c2 == 128) or Ls.i in home:
            return False
        cons = consumers.get(Ls.i, [])
        if len(cons) != 1 or cons[0] != Ls.i + 1:
            return False
        nxt = self.arch.layers[cons[0]]
        return nxt.kind == "C2f" and nxt.c1 == 128 and nxt.c2 == 128 and list(nxt.src) in ([-1], [Ls.i])

    def _plan_fork(self):
        """Round 4, CU partition inside the plan -- MEASURED AND LEFT OFF (MOY_FORK_VALUE=<units> switches it on).
        tools/probes/cu_share.py: the write-bound P3 value projection (9.9 GB per 288 frames) keeps 72 % of its rate on HALF of the
        compute units and a chain of eight P4-level 3x3 convs keeps 65 % of its rate on the other half, so side by side that pair
        finishes in 0.82 of its back-to-back time (0.95 as two whole-chip streams).  The plan itself has no such partner for the
        launch: between the moment layer 15 has produced P3 and the first deformable sampling there are 1.7 ms of mixed launches (two
        such convs; the rest 1x1 products that are bandwidth bound themselves and P5-level launches that do not fill even half a
        chip), and a pair shorter than ~1.4 ms of conv work measures >= 1.0 in the probe too.  In the engine: 41.00 ms (plan order)
        vs 41.07 ms (forked, 128 | 128 units) per 288-frame pass eagerly on one stream, 37.52 vs 38.32 ms per 2 x 288 frames from the
        hipGraphs of bench.py (112 and 144 units: 37.94, 38.49).  The fork stays as a switch with its bit-identity test; the two
        free-running engines of StreamedEngines remain the overlap mechanism."""
        self._fork = None
        # Round 6, small batches: the value projection of the classic plan (one launch over all B x S tokens, the longest of a small-batch
        # pass) depends on nothing between input_proj and the first deformable sampling, and the launches between them -- score pass, query
        # selection, box head, first self-attention -- are a chain of small kernels that leave the chip mostly idle: it runs beside them on
        # a side stream (two parallel branches of the hipGraph), whole chip for both (no CU limit).  Bit-identical by construction.
        vs = getattr(self, "_value_step", None)
        if vs is not None and 0 < self.B * self.S < int(self.opt.fork_small_value) and not int(self.opt.fork_value):
            join = next((i for i, m in enumerate(self.meta) if m["name"].startswith("msda")), None)
            if join is not None and 0 < vs < self._topk_step - 1 < join:
                # issued once the SCORE pass is through (itself a chip-filling launch: beside it the value launch only takes turns), so
                # that it runs beside the chain of small launches behind the score pass
                self._side_stream = torch.cuda.Stream(device=self.dev)
                self._fork = dict(side=vs, after=self._topk_step - 1, join=join, side_cus=0, main_cus=0)
                return
        L_ = int(self.opt.fork_value)
        step = getattr(self, "_value_p3_step", None)
        first = getattr(self, "_layer_first_step", {})
        if L_ <= 0 or step is None or not getattr(self, "fold_proj", False) or 16 not in first:
            return
        join = next((i for i, m in enumerate(self.meta) if m["name"].startswith("msda")), None)
        after = first[16] - 1
        if join is None or not (0 <= after < step < join):
            return
        ncu = torch.cuda.get_device_properties(self.dev).multi_processor_count
        L_ = max(8, min(ncu - 8, L_ * ncu // 256)) // 8 * 8
        self._side_stream = torch.cuda.Stream(device=self.dev)
        self._fork = dict(side=step, after=after, join=join, side_cus=L_, main_cus=ncu - L_)

    def _fold_plan(self, head_src, valid_host):
        """Conditions of the folded head (round 4) and its host-side constants, or None for the classic plan.  Folded when: a 16-bit engine
        at a batch where every per-level launch takes the weight-stationary kernel (B * h*w >= `fold_min_rows` = 8192 on the smallest
        level -- 13 frames of 640x1088; the 16-bit tiled kernel has neither the row runs nor the second row numbering), head-plane value layout, <= 4 classes (the fused narrow
        head), level widths of 128 / 256 channels, a valid mask that is one rectangle per level."""
        arch, B, sd = self.arch, self.B, self.sd
        is32 = self.dtype == torch.float32
        if not self.opt.fold_proj or self.opt.value_planes != 2:
            return None
        if not self.opt.score_runs or arch.nc > 4 or arch.hd != 256:
            return None
        if any(c not in (128, 256) for c in arch.head_ch):
            return None
        # 16-bit engines: every per-level launch must take the weight-stationary kernel (row runs, output row remap into head planes);
        # round 6, fp32 engines (exact and split-fp16): the TILED kernel has the row runs and the remap for fp32 tensors at any launch
        # size, so the fold holds at every batch -- what is asked below is only that a level's tensor fits one buffer descriptor
        # (round 6: the library takes the value / score forms of the weight-stationary kernel from 8192 rows -- `fold_min_rows`; the query
        #  below is what decides, this is only the cheap first look)
        if not is32 and (not self.opt.assume_wreg or any(B * h_ * w_ < int(self.opt.fold_min_rows) for h_, w_ in self.shapes)):
            return None
        if is32 and any(B * h_ * w_ * v.ld * 4 > 0x7fffffff for (v, _), (h_, w_) in zip(head_src, self.shapes)):
            return None
        if any((v.ld % 8) or (v.ptr % 16) for v, _ in head_src):
            return None
        d = f"model.{len(arch.layers)}.decoder"
        rect = valid_rectangles(valid_host, self.shapes)
        if rect is None or any(r_ is None for r_ in rect):
            return None
        s_, t_, sWp = [], [], []
        for li in range(len(self.shapes)):
            p = f"{d}.input_proj.{li}.1"
            scale = (sd[p + ".weight"] / torch.sqrt(sd[p + ".running_var"] + BN_EPS)).double()
            shift = sd[p + ".bias"].double() - sd[p + ".running_mean"].double() * scale
            Wp = sd[f"{d}.input_proj.{li}.0.weight"].reshape(arch.hd, arch.head_ch[li]).double()
            s_.append(scale); t_.append(shift); sWp.append(scale[:, None] * Wp)
        # the folded launches exist in the weight-stationary kernel only (head planes + output row remap; row runs with a
        # second A numbering) -- ask the library, on the host, whether it takes every one of them at this batch (byte-range limits of
        # its 32-bit lane offsets: a plane <= 1 GiB, M * lda * 2 <= 4 GiB, ...); anything else keeps the classic plan
        S, hd, dh = self.S, arch.hd, arch.hd // arch.nh
        al = self._dev(torch.zeros(64, dtype=torch.float32)).data_ptr()       # any 16-byte aligned device address: the query dereferences nothing
        off = 0
        for li, ((v, _), (h_, w_)) in enumerate(zip(head_src, self.shapes)):
            if is32:
                break
            y0, y1, x0, x1 = rect[li]
            a = L.GemmArgs()
            a.A, a.lda, a.W, a.M, a.N, a.K, a.ksize, a.stride = v.ptr, v.ld, al, B * h_ * w_, arch.ndl * hd, arch.head_ch[li], 1, 1
            a.shift, a.C, a.ldc, a.dtype = al, al, dh, self.code
            Sv_ = S - self.shapes[0][0] * self.shapes[0][1] if self._p3raw_wanted else S       # tokens per frame in the planes the plan will make
            a.plane_cols, a.plane_stride, a.c_rows_per_batch, a.c_batch_stride = dh, B * Sv_ * dh, h_ * w_, Sv_
            b = L.GemmArgs()
            b.A, b.lda, b.W, b.M, b.N, b.K, b.ksize, b.stride = v.ptr, v.ld, al, B * S, hd, arch.head_ch[li], 1, 1
            b.shift, b.ln_g, b.ln_b, b.dot_w, b.dot_b, b.dot_out, b.dot_n, b.dtype = al, al, al, al, al, al, arch.nc, self.code
            b.run_levels, b.run_period, b.run_a_period, b.run_a_off = 1, S, h_ * w_, off
            b.run_tok0[0], b.run_pitch[0], b.run_len[0], b.run_rows[0] = off + y0 * w_ + x0, w_, x1 - x0 + 1, y1 - y0 + 1
            for q_ in ((b,) if (li == 0 and self._p3raw_wanted) else (a, b)):     # (level 0 sampled raw: its value launch is never made)
                k = C.c_int(0)
                if self.lib.moy_gemm_query(C.byref(q_), C.byref(k)) != 0 or k.value != L.KERNEL_WREG:
                    return None
            off += h_ * w_
        return dict(rect=rect, s=s_, t=t_, sWp=sWp)

    def _score_runs(self, valid_host):
        """Round 3: the score pass over the VALID tokens only.  A masked token's enc_output feature is LN(enc_output.bias) whatever
        the frame shows (head.py:1039: `valid_mask * feats`), so its score is a constant: the pass just planned (all S tokens, masked
        rows zeroed) runs ONCE here and leaves that constant in the static `scores_all` buffer; the step then replaces it by the same
        launch restricted to the valid tokens, which `_generate_anchors` (head.py:1007) makes one rectangle per pyramid level --
        verified on the mask itself, any other pattern keeps the masked pass.  46 % fewer rows at 1088x608."""
        if self.dtype == torch.float32 or not self.opt.score_runs:
            return
        v = valid_host.bool()
        rects = valid_rectangles(v, self.shapes)
        if rects is None:
            return                                       # not a rectangle per level: keep the masked pass
        tok0, pitch, rlen, rows = [], [], [], []
        off = 0
        for (h_, w_), rc_ in zip(self.shapes, rects):
            if rc_ is not None:
                y0, y1, x0, x1 = rc_
                tok0.append(off + y0 * w_ + x0); pitch.append(w_); rlen.append(x1 - x0 + 1); rows.append(y1 - y0 + 1)
            off += h_ * w_
        if not tok0 or len(tok0) > 4 or sum(a * b for a, b in zip(rlen, rows)) != int(v.sum()):
            return
        fn, args = self._steps[-1]
        a_full = args[0]._obj                             # the GemmArgs of the masked pass (ctypes byref -> object)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        self.feats.buf.uniform_(-1.0, 1.0)                # (an intermediate buffer: every step overwrites it) so that the self-check below compares real scores
        L.check(fn(C.byref(a_full), st), "moy_gemm (score pass, all tokens)")      # constant scores of the masked tokens, once
        torch.cuda.synchronize()
        a = L.GemmArgs()
        C.memmove(C.byref(a), C.byref(a_full), C.sizeof(L.GemmArgs))
        a.a_mask, a.mask_period = None, 0
        a.run_levels, a.run_period = len(tok0), self.S
        for i in range(len(tok0)):
            a.run_tok0[i], a.run_pitch[i], a.run_len[i], a.run_rows[i] = tok0[i], pitch[i], rlen[i], rows[i]
        ref_scores = self.scores_all.clone()
        rc = fn(C.byref(a), st)
        torch.cuda.synchronize()
        if rc != 0:
            return                                       # MOY_ENOSYS: the shape does not take the weight-stationary score kernel
        assert torch.equal(ref_scores, self.scores_all), "score pass over the valid tokens differs from the pass over all tokens"
        self._keep.append(a)
        nv = sum(a_ * b_ for a_, b_ in zip(rlen, rows))
        self._steps[-1] = (fn, (C.byref(a),))
        m_ = self.meta[-1]
        frac = nv / self.S
        m_.update(name=m_["name"] + f" valid-runs {nv}/{self.S}", bytes=int(m_["bytes"] * frac), flops=int(m_["flops"] * frac))

    def reset_sequence(self, which=None):
        """Start of a new video sequence.  Temporal mode: empties the query memory and restarts the id counter of the
        sequences `which` (indices into the batch; default all) -- `is_first` of MOTRTrack.forward (head.py:199-205).
        side_state: FSQM.reset (fsqm.py:182-190)."""
        if self.n_max:
            t = self.trk
            idx = slice(None) if which is None else torch.as_tensor(which, device=self.dev, dtype=torch.long)
            t["n"][idx] = 0
            t["max_obj_id"][idx] = 0
            t["id"][idx] = -1
            t["dis"][idx] = 0
        if self.side_state:
            f = self.fsqm
            L.check(self.lib.moy_fsqm_reset(f["mem"].data_ptr(), f["conf"].data_ptr(), f["ids"].data_ptr(), f["boxes"].data_ptr(),
                                            f["low"].data_ptr(), f["pool"].data_ptr(), f["pool"].numel(), f["pool_hc"].data_ptr(),
                                            C.c_void_p(torch.cuda.current_stream().cuda_stream)), "moy_fsqm_reset")

    def _linear_w_raw(self, w, b):
        return self._weight(w), self._dev(b)

    def _ln(self, p):
        return (self._dev(self.sd[p + ".weight"]), self._dev(self.sd[p + ".bias"]))

    # ------------------------------------------------------------------ run
    def run_steps(self, start=0, stop=None, slot=0):
        """Enqueue launches [start, stop) on the current stream, reading the frames from input slot `slot`.
        A call that covers the whole forked segment (`self._fork`, see `_plan_fork`) issues the side launch on the engine's side
        stream between two events -- also under graph capture, where the pair becomes two parallel branches of the hipGraph; a call
        over a part of the plan (per-launch timing, forward_with_topk) runs every launch in plan order on the whole chip."""
        cur = torch.cuda.current_stream()
        st = C.c_void_p(cur.cuda_stream)
        stem = getattr(self, "_stem_step", -1) if slot else -1
        fk = getattr(self, "_fork", None)
        n = len(self._steps)
        if fk is not None and not (start <= fk["after"] and (stop is None or stop >= n or stop > fk["join"])):
            fk = None
        try:
            for i, (fn, args) in enumerate(self._steps[start:stop], start):
                if fk is not None:
                    if i == fk["side"]:
                        continue                                   # issued at the fork point
                    if i == fk["join"]:
                        cur.wait_stream(self._side_stream)
                        self.lib.moy_set_cu_limit(0)
                if i == stem:
                    args = (self.inputs[slot].data_ptr(),) + args[1:]
                rc = fn(*args, st)
                if rc != 0:
                    L.check(rc, fn.__name__)
                if fk is not None and i == fk["after"]:
                    self._side_stream.wait_stream(cur)
                    self.lib.moy_set_cu_limit(fk["side_cus"])                 # (grid size of the persistent kernels, this thread's next launches)
                    sfn, sargs = self._steps[fk["side"]]
                    rc = sfn(*sargs, C.c_void_p(self._side_stream.cuda_stream))
                    if rc != 0:
                        L.check(rc, sfn.__name__)
                    self.lib.moy_set_cu_limit(fk["main_cus"])
        finally:
            if fk is not None:
                self.lib.moy_set_cu_limit(0)

    def forward(self, frames: torch.Tensor | None = None, slot: int = 0):
        """frames: uint8 [B,H,W,3] BGR (input_format 'u8') or float32 [B,3,H,W] RGB in [0,1] ('f32'),
        already on the device; copied into input slot `slot`.  frames=None: the slot already holds them (a producer wrote
        `self.inputs[slot]` in place) -- no copy at all.  Returns `outputs()`."""
        if frames is not None:
            self.inputs[slot].copy_(frames, non_blocking=True)
        g = self._graphs.get(slot)
        if g is not None:
            g.replay()
        else:
            self.run_steps(slot=slot)
        return self.outputs()

    def forward_head(self, feats):
        """MOTRTrack.forward surface: feats = [P3, P4, P5] NCHW-shaped device tensors."""
        for v, (h_, w_), f in zip(self.head_inputs, self.shapes, feats):
            v.buf.view(self.B, h_, w_, -1).copy_(f.permute(0, 2, 3, 1))
        self.run_steps(self._head_start)
        return self.outputs()

    def forward_with_topk(self, frames, topk_local: torch.Tensor):
        """Test hook: run with an injected query selection [B, nq] (int) instead of moy_topk."""
        self.input.copy_(frames)
        self.run_steps(0, self._topk_step)
        tl = topk_local.to(self.dev, torch.int32).reshape(self.B, -1)
        self.topk_local.copy_(tl)
        self.topk_global.copy_(tl + torch.arange(self.B, device=self.dev, dtype=torch.int32)[:, None] * self.S)
        self.n_masked.zero_()
        self.run_steps(self._topk_step + 1)
        return self.outputs()

    def capture(self, warmup: int = 2):
        """Capture the step into a hipGraph per input slot (HIP streams + graphs instead of a tracing compiler)."""
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self.run_steps()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        head_only = self.input is None
        for slot in range(1 if head_only else self.n_inputs):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self.run_steps(slot=slot)
            self._graphs[slot] = g
        self._graph = self._graphs[0]
        return self._graph

    def capture_split(self, warmup: int = 2):
        """Two hipGraphs: launches [0, _split) (backbone, projections and score pass over all tokens) and [_split, end) (query
        selection + decoder: a dependent chain of query-sized launches).  StreamedEngines replays the second on a high-priority
        stream so that its small launches are dispatched ahead of the other sub-batch's full-chip kernels."""
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self.run_steps()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g0, g1 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(g0):
            self.run_steps(0, self._split)
        with torch.cuda.graph(g1):
            self.run_steps(self._split)
        self._graph_lo, self._graph_hi = g0, g1
        return g0, g1

    def outputs(self):
        if self.arch.head_kind == "detect":
            return dict(y=self.y, rows=self.rows, n_rows=self.n_rows)
        B, nq = self.B, self.arch.nq
        Lq = nq + self.n_max             # temporal mode: rows = [track slots | detect queries] per sequence
        out = dict(y=self.y, scores=self.scores, obj_idxes=self.obj_idxes, rows=self.rows, track_id=self.track_id,
                   n_rows=self.n_rows, n_ids=self.n_ids, logits=self.logits.view(B, Lq, -1),
                   boxes=self.boxes.view(B, Lq, 4), hs=self.hs.tensor().view(B, Lq, -1),
                   topk_ind=self.topk_local, n_masked=self.n_masked, refer_bbox_logit=self.refer_logit.view(B, nq, 4))
        if self.n_max:
            t = self.trk
            out.update(n_tracks=t["n"], n_overflow=self.n_overflow, trk_id=t["id"], trk_dis=t["dis"],
                       trk_ref=t["ref"].view(B, self.n_max, 4), trk_qpos=t["qpos"].view(B, self.n_max, -1),
                       trk_embed=t["embed"].view(B, self.n_max, -1), max_obj_id=t["max_obj_id"])
        return out

    @property
    def num_launches(self):
        return len(self._steps)


class StreamedEngines:
    """A step over `batch` frames cut into `streams` sub-batches, each with its own TrackEngine (static buffers), HIP
    stream and hipGraph.  The decoder is a chain of small, latency-bound launches (M = 300 rows per frame); on its own
    stream it runs beside the bandwidth-bound backbone launches of the other sub-batch instead of leaving the chip idle
    (measured: 2 streams x 96 frames 8.8k FPS vs 1 x 96 8.2k, same device).  Frames stay independent (per-frame reset
    semantics), so the split is result-neutral (tests/test_gpu_engine.py)."""

    def __init__(self, arch, sd, H, W, batch, streams=2, graph=True, split_priority=False, **kw):
        if batch % streams:
            raise ValueError("batch must be a multiple of streams")
        self.B, self.S, self.Bs = batch, streams, batch // streams
        self.engines = [TrackEngine(arch, sd, H, W, batch=self.Bs, **kw) for _ in range(streams)]
        dev = self.engines[0].dev
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(streams)]
        # split_priority: the query-sized chain of every sub-batch replays from its own hipGraph on a high-priority stream
        self._split = bool(split_priority and graph and hasattr(self.engines[0], "_split"))
        self.hi_streams = [torch.cuda.Stream(device=dev, priority=-1) for _ in range(streams)] if self._split else []
        self._graph = graph
        self._warm = False

    def load(self, frames, slot=0):
        """Write a batch [batch, ...] of device-resident frames into input slot `slot` of the sub-batch engines (the producer
        side of the slot protocol: done once per batch, outside the step)."""
        for k, e in enumerate(self.engines):
            e.inputs[slot].copy_(frames[k * self.Bs:(k + 1) * self.Bs], non_blocking=True)

    def _warmup(self, frames, slot=0):
        for k, e in enumerate(self.engines):
            e.forward(frames[k * self.Bs:(k + 1) * self.Bs] if frames is not None else None, slot=slot)
        torch.cuda.synchronize()
        if self._graph:
            for e in self.engines:
                if self._split:
                    e.capture_split()
                else:
                    e.capture()
        self._warm = True

    def forward(self, frames=None, slot=0):
        """frames [batch, ...] resident on the device (copied into input slot `slot`), or None when `load(frames, slot)` /
        an in-place producer already filled the slot: the step then moves no frame bytes.  Enqueue-only: outputs are valid
        after `synchronize()`."""
        if self._split and slot != 0:
            # capture_split() records its two graphs for input slot 0 only (the slot used to be ignored silently)
            raise ValueError("split_priority replays the slot-0 graphs: use slot 0, or capture() per slot without split_priority")
        if not self._warm:
            self._warmup(frames, slot)
        cur = torch.cuda.current_stream()
        for k, (e, st) in enumerate(zip(self.engines, self.streams)):
            st.wait_stream(cur)                                   # the frames were produced on the caller's stream
            sub = frames[k * self.Bs:(k + 1) * self.Bs] if frames is not None else None
            if self._split:
                hi = self.hi_streams[k]
                st.wait_stream(hi)                                # the previous step's chain still reads this engine's buffers
                with torch.cuda.stream(st):
                    if sub is not None:
                        e.input.copy_(sub, non_blocking=True)
                    e._graph_lo.replay()
                hi.wait_stream(st)
                with torch.cuda.stream(hi):
                    e._graph_hi.replay()
            else:
                with torch.cuda.stream(st):
                    e.forward(sub, slot=slot)
        return [e.outputs() for e in self.engines]

    def synchronize(self):
        for st in self.streams + self.hi_streams:
            st.synchronize()

    def outputs(self):
        """Per-frame outputs of the whole batch, concatenated over the sub-batches (after `synchronize()`)."""
        outs = [e.outputs() for e in self.engines]
        return {k: torch.cat([o[k] for o in outs], 0) for k in outs[0]}


def valid_rectangles(valid, shapes):
    """The valid-token mask of `_generate_anchors` (head.py:1007), level by level, as rectangles (y0, y1, x0, x1) inclusive -- or None
    for a level without valid tokens; the whole result is None when some level's mask is NOT one rectangle (the callers then keep
    the masked pass over all tokens).  valid: bool / uint8 [S] in level-major order, shapes: [(h, w), ...]."""
    v = torch.as_tensor(valid).bool().flatten()
    out, off = [], 0
    for (h_, w_) in shapes:
        m = v[off:off + h_ * w_].view(h_, w_)
        off += h_ * w_
        ys, xs = m.any(1).nonzero().flatten(), m.any(0).nonzero().flatten()
        if len(ys) == 0:
            out.append(None)
            continue
        y0, y1, x0, x1 = int(ys[0]), int(ys[-1]), int(xs[0]), int(xs[-1])
        rect = torch.zeros_like(m)
        rect[y0:y1 + 1, x0:x1 + 1] = True
        if not torch.equal(rect, m):
            return None
        out.append((y0, y1, x0, x1))
    return out


def _generate_anchors(shapes, grid_size=0.05, eps=1e-2):
    """Input-independent anchors + validity mask, computed once on the host with the formula the
    reference ships (nn/modules/head.py:993-1010) INCLUDING its swapped normalisation
    cx=(x+.5)/H, cy=(y+.5)/W: the mask is part of the parity target."""
    anchors = []
    for i, (h, w) in enumerate(shapes):
        gy, gx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
        xy = (torch.stack([gx, gy], -1).unsqueeze(0) + 0.5) / torch.tensor([h, w], dtype=torch.float32)
        wh = torch.ones_like(xy) * grid_size * (2.0 ** i)
        anchors.append(torch.cat([xy, wh], -1).view(-1, h * w, 4))
    a = torch.cat(anchors, 1)
    valid = ((a > eps) * (a < 1 - eps)).all(-1, keepdim=True)
    a = torch.log(a / (1 - a)).masked_fill(~valid, float("inf"))
    return a, valid

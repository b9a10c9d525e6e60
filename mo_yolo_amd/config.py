"""Architecture description of the tracking graph (own restatement, no reference import).

Mirrors what `parse_model` (ultralytics/nn/tasks.py:881-956) produces from
`ultralytics/cfg/models/v8/yolo_track.yaml:15-46` for a given (depth_multiple,
width_multiple): channel widths are `ceil(c*width/8)*8` (utils/ops.py:132), repeats
`max(round(n*depth),1)` for n>1, the head reads layers (15, 18, 21).

`param_shapes()` enumerates the reference `state_dict` keys/shapes (SURVEY §5
"Checkpoint / resume": key names = reference keys) so weight files interchange.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import List, Tuple

# (from, repeats, kind, args) -- yolo_track.yaml:15-46
_YOLO_TRACK = [
    (-1, 1, "Conv", (64, 3, 2)),      # 0  P1/2
    (-1, 1, "Conv", (128, 3, 2)),     # 1  P2/4
    (-1, 3, "C2f", (128, True)),      # 2
    (-1, 1, "Conv", (256, 3, 2)),     # 3  P3/8
    (-1, 6, "C2f", (256, True)),      # 4
    (-1, 1, "Conv", (512, 3, 2)),     # 5  P4/16
    (-1, 6, "C2f", (512, True)),      # 6
    (-1, 1, "Conv", (512, 3, 2)),     # 7  P5/32
    (-1, 3, "C2f", (512, True)),      # 8
    (-1, 1, "SPPF", (512, 5)),        # 9
    (-1, 1, "Upsample", ()),          # 10
    ((-1, 6), 1, "Concat", ()),       # 11
    (-1, 3, "C2f", (512, False)),     # 12
    (-1, 1, "Upsample", ()),          # 13
    ((-1, 4), 1, "Concat", ()),       # 14
    (-1, 3, "C2f", (256, False)),     # 15 -> head P3
    (-1, 1, "Conv", (256, 3, 2)),     # 16
    ((-1, 12), 1, "Concat", ()),      # 17
    (-1, 3, "C2f", (512, False)),     # 18 -> head P4
    (-1, 1, "Conv", (512, 3, 2)),     # 19
    ((-1, 9), 1, "Concat", ()),       # 20
    (-1, 3, "C2f", (512, False)),     # 21 -> head P5
]
HEAD_FROM = (15, 18, 21)

# (from, repeats, kind, args) -- ultralytics/cfg/models/v8/yolov8.yaml:15-46 (config C1: YOLOv8n detect)
_YOLO_V8 = [
    (-1, 1, "Conv", (64, 3, 2)), (-1, 1, "Conv", (128, 3, 2)), (-1, 3, "C2f", (128, True)),
    (-1, 1, "Conv", (256, 3, 2)), (-1, 6, "C2f", (256, True)), (-1, 1, "Conv", (512, 3, 2)),
    (-1, 6, "C2f", (512, True)), (-1, 1, "Conv", (1024, 3, 2)), (-1, 3, "C2f", (1024, True)),
    (-1, 1, "SPPF", (1024, 5)),
    (-1, 1, "Upsample", ()), ((-1, 6), 1, "Concat", ()), (-1, 3, "C2f", (512, False)),
    (-1, 1, "Upsample", ()), ((-1, 4), 1, "Concat", ()), (-1, 3, "C2f", (256, False)),
    (-1, 1, "Conv", (256, 3, 2)), ((-1, 12), 1, "Concat", ()), (-1, 3, "C2f", (512, False)),
    (-1, 1, "Conv", (512, 3, 2)), ((-1, 9), 1, "Concat", ()), (-1, 3, "C2f", (1024, False)),
]


def make_divisible(x, d=8):
    return int(math.ceil(x / d) * d)


@dataclass
class LayerSpec:
    i: int
    kind: str            # Conv | C2f | SPPF | Upsample | Concat
    src: Tuple[int, ...]  # absolute indices of the producing layers (-1 => previous)
    c1: int
    c2: int
    k: int = 1
    s: int = 1
    n: int = 1           # C2f repeats
    shortcut: bool = False


@dataclass
class TrackArch:
    depth: float = 0.33
    width: float = 0.50
    nc: int = 1
    nq: int = 300
    hd: int = 256        # MYDecoder defaults, head.py:810-828
    nh: int = 8
    ndl: int = 6
    ndp: int = 4
    d_ffn: int = 1024
    layers: List[LayerSpec] = field(default_factory=list)
    head_ch: Tuple[int, ...] = ()
    head_kind: str = "track"     # "track" (MOTRTrack, head.py:90) | "detect" (Detect, head.py:27)

    @property
    def nl(self):
        return len(self.head_ch)


def build_detect_arch(depth=0.33, width=0.25, nc=80, max_channels=1024) -> TrackArch:
    """YOLOv8 detection graph at a `scales` entry of yolov8.yaml (default 'n': 0.33 / 0.25 / 1024)."""
    return build_arch(depth, width, nc, nq=0, table=_YOLO_V8, max_channels=max_channels, head_kind="detect")


def build_arch(depth=0.33, width=0.50, nc=1, nq=300, table=None, max_channels=float("inf"), head_kind="track") -> TrackArch:
    arch = TrackArch(depth=depth, width=width, nc=nc, nq=nq, head_kind=head_kind)
    ch: List[int] = []
    c_prev = 3
    for i, (f, n, kind, args) in enumerate(table or _YOLO_TRACK):
        n = max(round(n * depth), 1) if n > 1 else n
        src = tuple((i - 1 if j == -1 else j) for j in ((f,) if isinstance(f, int) else f))
        c1 = c_prev if src[0] == i - 1 and i > 0 else (3 if i == 0 else ch[src[0]])
        if kind == "Conv":
            c2 = make_divisible(min(args[0], max_channels) * width)      # tasks.py:911
            spec = LayerSpec(i, kind, src, c1, c2, k=args[1], s=args[2])
        elif kind == "C2f":
            c2 = make_divisible(min(args[0], max_channels) * width)
            spec = LayerSpec(i, kind, src, c1, c2, n=n, shortcut=args[1])
        elif kind == "SPPF":
            c2 = make_divisible(min(args[0], max_channels) * width)
            spec = LayerSpec(i, kind, src, c1, c2, k=args[1])
        elif kind == "Upsample":
            spec = LayerSpec(i, kind, src, c1, c1)
        elif kind == "Concat":
            c2 = sum(ch[j] for j in src)
            spec = LayerSpec(i, kind, src, c2, c2)
        else:
            raise ValueError(kind)
        arch.layers.append(spec)
        ch.append(spec.c2)
        c_prev = spec.c2
    arch.head_ch = tuple(ch[j] for j in HEAD_FROM)
    return arch


def level_shapes(H, W, strides=(8, 16, 32)):
    """Feature-map (h, w) per head level for a network input H x W (H, W % 32 == 0)."""
    assert H % 32 == 0 and W % 32 == 0, "network input must be a multiple of 32 (data/loaders.py:316-332)"
    return [(H // s, W // s) for s in strides]


def _conv_bn(sd, p, c1, c2, k):
    sd[p + ".conv.weight"] = (c2, c1, k, k)
    _bn(sd, p + ".bn", c2)


def _bn(sd, p, c):
    sd[p + ".weight"] = (c,)
    sd[p + ".bias"] = (c,)
    sd[p + ".running_mean"] = (c,)
    sd[p + ".running_var"] = (c,)
    sd[p + ".num_batches_tracked"] = ()


def _linear(sd, p, cin, cout):
    sd[p + ".weight"] = (cout, cin)
    sd[p + ".bias"] = (cout,)


def _ln(sd, p, c):
    sd[p + ".weight"] = (c,)
    sd[p + ".bias"] = (c,)


def _mha(sd, p, c):
    sd[p + ".in_proj_weight"] = (3 * c, c)
    sd[p + ".in_proj_bias"] = (3 * c,)
    _linear(sd, p + ".out_proj", c, c)


def param_shapes(arch: TrackArch) -> "OrderedDict[str, tuple]":
    """Reference state_dict keys -> shapes, in the reference's registration order."""
    sd: "OrderedDict[str, tuple]" = OrderedDict()
    for L in arch.layers:
        p = f"model.{L.i}"
        if L.kind == "Conv":
            _conv_bn(sd, p, L.c1, L.c2, L.k)
        elif L.kind == "C2f":
            c = L.c2 // 2                                   # block.py:173  int(c2*0.5)
            _conv_bn(sd, p + ".cv1", L.c1, 2 * c, 1)
            _conv_bn(sd, p + ".cv2", (2 + L.n) * c, L.c2, 1)
            for j in range(L.n):
                _conv_bn(sd, f"{p}.m.{j}.cv1", c, c, 3)     # Bottleneck e=1.0, block.py:175
                _conv_bn(sd, f"{p}.m.{j}.cv2", c, c, 3)
        elif L.kind == "SPPF":
            c_ = L.c1 // 2
            _conv_bn(sd, p + ".cv1", L.c1, c_, 1)
            _conv_bn(sd, p + ".cv2", c_ * 4, L.c2, 1)
    h = f"model.{len(arch.layers)}"
    hd, nc = arch.hd, arch.nc
    if arch.head_kind == "detect":                            # Detect.__init__, head.py:35-47
        c2, c3 = max(16, arch.head_ch[0] // 4, 64), max(arch.head_ch[0], min(nc, 100))
        for name, cm, cout in (("cv2", c2, 64), ("cv3", c3, nc)):
            for li, c in enumerate(arch.head_ch):
                _conv_bn(sd, f"{h}.{name}.{li}.0", c, cm, 3)
                _conv_bn(sd, f"{h}.{name}.{li}.1", cm, cm, 3)
                sd[f"{h}.{name}.{li}.2.weight"] = (cout, cm, 1, 1)
                sd[f"{h}.{name}.{li}.2.bias"] = (cout,)
        sd[h + ".dfl.conv.weight"] = (1, 16, 1, 1)
        return sd
    d = h + ".decoder"
    for li, c in enumerate(arch.head_ch):
        sd[f"{d}.input_proj.{li}.0.weight"] = (hd, c, 1, 1)
        _bn(sd, f"{d}.input_proj.{li}.1", hd)
    for li in range(arch.ndl):
        q = f"{d}.decoder.layers.{li}"
        _mha(sd, q + ".self_attn", hd)
        _ln(sd, q + ".norm1", hd)
        _linear(sd, q + ".cross_attn.sampling_offsets", hd, arch.nh * arch.nl * arch.ndp * 2)
        _linear(sd, q + ".cross_attn.attention_weights", hd, arch.nh * arch.nl * arch.ndp)
        _linear(sd, q + ".cross_attn.value_proj", hd, hd)
        _linear(sd, q + ".cross_attn.output_proj", hd, hd)
        _ln(sd, q + ".norm2", hd)
        _linear(sd, q + ".linear1", hd, arch.d_ffn)
        _linear(sd, q + ".linear2", arch.d_ffn, hd)
        _ln(sd, q + ".norm3", hd)
    sd[d + ".denoising_class_embed.weight"] = (nc, hd)
    _linear(sd, d + ".query_pos_head.layers.0", 4, 2 * hd)
    _linear(sd, d + ".query_pos_head.layers.1", 2 * hd, hd)
    _linear(sd, d + ".enc_output.0", hd, hd)
    _ln(sd, d + ".enc_output.1", hd)
    _linear(sd, d + ".enc_score_head", hd, nc)
    for j, (a, b) in enumerate(((hd, hd), (hd, hd), (hd, 4))):
        _linear(sd, f"{d}.enc_bbox_head.layers.{j}", a, b)
    for li in range(arch.ndl):
        _linear(sd, f"{d}.dec_score_head.{li}", hd, nc)
    for li in range(arch.ndl):
        for j, (a, b) in enumerate(((hd, hd), (hd, hd), (hd, 4))):
            _linear(sd, f"{d}.dec_bbox_head.{li}.layers.{j}", a, b)
    t = h + ".track_embed"                                   # QIM, qim.py:83-115
    _mha(sd, t + ".self_attn", hd)
    _linear(sd, t + ".linear1", hd, hd)
    _linear(sd, t + ".linear2", hd, hd)
    _linear(sd, t + ".linear_feat1", hd, hd)
    _linear(sd, t + ".linear_feat2", hd, hd)
    _ln(sd, t + ".norm_feat", hd)
    _ln(sd, t + ".norm1", hd)
    _ln(sd, t + ".norm2", hd)
    return sd

"""ctypes binding of libmoyolo.so (the C ABI declared in include/moyolo.h).

There is NO fallback: if the shared library is missing or a symbol is absent the import of any
compute entry point raises.  (The CPU oracle lives in /oracle and is test infrastructure only.)
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# MOYOLO_LIB names another build of the same C ABI: the LAB library (libmoyolo_diag.so: the sources with -DMOY_DIAG=1 -- the `MOY_*`
# A/B knobs and the timing-only kernel instances live there and only there), or an older build for a same-device A/B
LIB_PATH = os.environ.get("MOYOLO_LIB") or os.path.join(HERE, "libmoyolo.so")
LAB_LIB_PATH = os.path.join(HERE, "libmoyolo_diag.so")


def lab_library() -> str:
    """Path of the lab build (`python -m mo_yolo_amd.build --diag`) for a child process's MOYOLO_LIB; raises if it has not been built."""
    if not os.path.exists(LAB_LIB_PATH):
        raise MoyoloError(f"{LAB_LIB_PATH} is missing: build it with `python -m mo_yolo_amd.build --diag`")
    return LAB_LIB_PATH

F32, BF16, F16 = 0, 1, 2
F32X3 = 3        # moy_gemm only: fp32 tensors, split-fp16 matrix arithmetic (include/moyolo.h: MOY_F32X3)
ACT_NONE, ACT_SILU, ACT_RELU, ACT_SIGMOID = 0, 1, 2, 3

vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float


class GemmArgs(C.Structure):
    """Mirror of `moy_gemm_args` (include/moyolo.h)."""
    _fields_ = [
        ("A", vp), ("A2", vp), ("a_rows", vp), ("a_rows_bound", i32), ("a_mask", vp), ("mask_period", i32), ("lda", i64),
        ("W", vp), ("M", i32), ("N", i32), ("K", i32), ("ksize", i32), ("stride", i32),
        ("B", i32), ("Hin", i32), ("Win", i32), ("Hout", i32), ("Wout", i32), ("Cin", i32),
        ("scale", vp), ("shift", vp), ("act", i32), ("R", vp), ("ldr", i64), ("ln_g", vp), ("ln_b", vp),
        ("C", vp), ("ldc", i64), ("out_f32", i32), ("dtype", i32),
        ("c_rows_per_batch", i32), ("c_batch_stride", i32),
        ("dot_w", vp), ("dot_b", vp), ("dot_out", vp), ("dot_n", i32),
        ("pre", vp), ("ld_pre", i64), ("pre_h", i32), ("pre_w", i32), ("a2_cols", i32),
        ("plane_cols", i32), ("plane_stride", i64),
        ("run_levels", i32), ("run_period", i32), ("run_tok0", i32 * 4), ("run_pitch", i32 * 4), ("run_len", i32 * 4), ("run_rows", i32 * 4),
        ("run_a_period", i32), ("run_a_off", i32),
        ("post_W", vp), ("post_scale", vp), ("post_shift", vp), ("post_n", i32), ("post_act", i32),
    ]


class DecoderTailArgs(C.Structure):
    """Mirror of `moy_decoder_tail_args` (include/moyolo.h)."""
    _fields_ = [
        ("samp", vp), ("ld_samp", i64), ("e1", vp), ("ld_e1", i64), ("M", i32),
        ("Wp", vp), ("bp", vp), ("ln2_g", vp), ("ln2_b", vp),
        ("W1", vp), ("b1", vp), ("W2", vp), ("b2", vp), ("d_ffn", i32),
        ("ln3_g", vp), ("ln3_b", vp), ("out", vp), ("ld_out", i64),
        ("B0", vp), ("c0", vp), ("B1", vp), ("c1", vp), ("w2", vp), ("c2", vp),
        ("ref_in", vp), ("ref_out", vp), ("dtype", i32),
        ("qpos", vp), ("ld_qpos", i64), ("out_xp", vp), ("ld_xp", i64), ("w_packed", i32),
        ("Wqkv", vp), ("bqkv", vp), ("qkv", vp), ("ld_qkv", i64),
    ]


class DecoderMidArgs(C.Structure):
    """Mirror of `moy_decoder_mid_args` (include/moyolo.h)."""
    _fields_ = [
        ("attn", vp), ("ld_attn", i64), ("x", vp), ("ld_x", i64), ("qpos", vp), ("ld_qpos", i64), ("M", i32),
        ("Wo", vp), ("bo", vp), ("ln_g", vp), ("ln_b", vp), ("Woa", vp), ("boa", vp), ("n_oa", i32),
        ("e1", vp), ("ld_e1", i64), ("offaw", vp), ("ld_oa", i64), ("dtype", i32), ("w_packed", i32),
    ]


class MsdaRawArgs(C.Structure):
    """Mirror of `moy_msda_raw_args` (include/moyolo.h)."""
    _fields_ = [
        ("x0", vp), ("ld0", i64), ("wc", vp), ("bc", vp), ("planes", vp), ("head_stride", i64), ("S1", i32),
        ("B", i32), ("Lq", i32), ("L", i32), ("shapes_hw", vp), ("offaw", vp), ("ld_oa", i64), ("ref", vp),
        ("out", vp), ("ldo", i64), ("dtype", i32), ("wc_packed", i32), ("perm", vp),
    ]


class C2fArgs(C.Structure):
    """Mirror of `moy_c2f_args` (include/moyolo.h)."""
    _fields_ = [
        ("x", vp), ("ldx", i64), ("B", i32), ("H", i32), ("W", i32),
        ("w_cv1", vp), ("kp_cv1", i32), ("scale_cv1", vp), ("shift_cv1", vp),
        ("w_m1", vp), ("scale_m1", vp), ("shift_m1", vp),
        ("w_m2", vp), ("scale_m2", vp), ("shift_m2", vp), ("kp_m", i32),
        ("w_cv2", vp), ("kp_cv2", i32), ("scale_cv2", vp), ("shift_cv2", vp),
        ("out", vp), ("ldo", i64), ("dtype", i32),
    ]


# name -> (restype, argtypes); every symbol declared in include/moyolo.h
SIGNATURES = {
    "moy_version": (C.c_int, []),
    "moy_strerror": (C.c_char_p, [C.c_int]),
    "moy_gemm": (C.c_int, [C.POINTER(GemmArgs), vp]),
    "moy_gemm_query": (C.c_int, [C.POINTER(GemmArgs), C.POINTER(C.c_int)]),
    "moy_stem_conv": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_int, vp, i64, C.c_int, vp]),
    "moy_stem_conv_mfma": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_int, vp, i64, vp]),
    "moy_stem_conv_x3": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_int, vp, i64, vp]),
    "moy_stem_l1_fused": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, i64, C.c_int, vp]),
    "moy_c2f_fused": (C.c_int, [C.POINTER(C2fArgs), vp]),
    "moy_sppf_pool": (C.c_int, [vp, i64, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, i64, C.c_int, vp]),
    "moy_upsample2x": (C.c_int, [vp, i64, C.c_int, C.c_int, C.c_int, C.c_int, vp, i64, C.c_int, vp]),
    "moy_rowdot": (C.c_int, [vp, i64, vp, C.c_int, C.c_int, vp, vp, C.c_int, C.c_int, vp, vp, vp, C.c_int, vp]),
    "moy_mlp_head": (C.c_int, [vp, i64, vp, C.c_int, vp, vp, vp, vp, vp, vp, C.c_int, vp, vp, vp, C.c_int, vp]),
    "moy_decoder_tail": (C.c_int, [C.POINTER(DecoderTailArgs), vp]),
    "moy_decoder_mid": (C.c_int, [C.POINTER(DecoderMidArgs), vp]),
    "moy_topk": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp]),
    "moy_pos2posemb": (C.c_int, [vp, C.c_int, vp, i64, C.c_int, vp]),
    "moy_mha_core": (C.c_int, [vp, i64, C.c_int, C.c_int, C.c_int, C.c_int, vp, i64, C.c_int, vp]),
    "moy_mha_core_masked": (C.c_int, [vp, i64, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, i64, C.c_int, vp]),
    "moy_temporal_assemble": (C.c_int, [vp, vp, vp, vp, vp, i64, vp, i64, vp, C.c_int, C.c_int, C.c_int, vp, i64, vp, i64, vp, vp,
                                        C.c_int, vp]),
    "moy_temporal_assign": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, f32, f32, C.c_int, f32, f32, f32,
                                      vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "moy_temporal_commit": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp]),
    "moy_msda_fused": (C.c_int, [vp, i64, i64, C.c_int, C.c_int, vp, C.c_int, vp, i64, vp, C.c_int, vp, i64, C.c_int, vp]),
    "moy_msda_raw0": (C.c_int, [C.POINTER(MsdaRawArgs), vp]),
    "moy_msda_fwd_f32": (C.c_int, [vp, vp, vp, vp, vp] + [C.c_int] * 7 + [vp, vp]),
    "moy_msda_fwd_bf16": (C.c_int, [vp, vp, vp, vp, vp] + [C.c_int] * 7 + [vp, vp]),
    "moy_msda_fwd_f16": (C.c_int, [vp, vp, vp, vp, vp] + [C.c_int] * 7 + [vp, vp]),
    "moy_msda_fwd_f64": (C.c_int, [vp, vp, vp, vp, vp] + [C.c_int] * 7 + [vp, vp]),
    "moy_msda_bwd_f32": (C.c_int, [vp, vp, vp, vp, vp, vp] + [C.c_int] * 7 + [vp, vp, vp, vp]),
    "moy_msda_bwd_f64": (C.c_int, [vp, vp, vp, vp, vp, vp] + [C.c_int] * 7 + [vp, vp, vp, vp]),
    "moy_resize_linear_u8": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, i64, i64, vp, C.c_int, C.c_int, vp]),
    "moy_box_iou": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]),
    "moy_assign_post": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, f32, f32, f32, f32, vp, vp, vp, vp, vp, vp, vp, vp]),
    "moy_track_state_update": (C.c_int, [vp, vp, vp, vp, i64, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, C.c_int, vp]),
    "moy_fsqm_reset": (C.c_int, [vp, vp, vp, vp, vp, vp, i32, vp, vp]),
    "moy_detect_decode": (C.c_int, [vp, i64, vp, i64, C.c_int, C.c_int, C.c_int, C.c_int, f32, C.c_int, C.c_int, vp, C.c_int, vp]),
    "moy_nms": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, f32, f32, C.c_int, f32, f32, f32, f32, f32, f32, vp, vp, vp]),
    "moy_gather_rows": (C.c_int, [vp, i64, vp, C.c_int, C.c_int, vp, i64, C.c_int, vp]),
    "moy_level_rows": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]),
    "moy_level_select": (C.c_int, [vp, i64, i64, vp, vp, vp, vp, C.c_int, C.c_int, vp, i64, C.c_int, vp]),
    "moy_set_cu_limit": (C.c_int, [C.c_int]),
    "moy_cast_f32_to": (C.c_int, [vp, i64, C.c_int, C.c_int, vp, i64, C.c_int, vp]),
    "moy_sigmoid_f32": (C.c_int, [vp, C.c_int, vp, vp]),
    "moy_query_order": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    "moy_msda_prep": (C.c_int, [vp, i64, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, vp, C.c_int, vp]),
    "moy_mask_rows": (C.c_int, [vp, i64, C.c_int, C.c_int, vp, C.c_int, vp]),
}

_lib = None


class MoyoloError(RuntimeError):
    pass


def lib():
    """Load libmoyolo.so (once).  Raises if it has not been built -- no silent fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MoyoloError(
                f"{LIB_PATH} is missing: build it with `python -m mo_yolo_amd.build` (hipcc, gfx950). "
                "mo_yolo_amd has no CPU or eager fallback for its compute path.")
        # PyTorch-ROCm ships its own libamdhip64: it must be the HIP runtime of the process BEFORE this library (linked against
        # the same soname) is loaded, otherwise two runtimes coexist and torch's streams / pointers are foreign to our launches
        # ("HIP launch error" on the first kernel; seen when build() loaded the library ahead of `import torch`).
        import torch  # noqa: F401
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(l, name)      # AttributeError if the ABI drifted
            except AttributeError:
                # an OLDER build named by MOYOLO_LIB for a same-device A/B (tools/stress_rings.py): entry points added since are
                # simply absent there -- announced, and any call of one still fails.  The library of the tree itself must be complete.
                if os.environ.get("MOYOLO_LIB") and os.environ.get("MOYOLO_LIB_ALLOW_MISSING") == "1":
                    import sys
                    print(f"[mo_yolo_amd] {LIB_PATH} has no {name} (older A/B build)", file=sys.stderr)
                    continue
                raise
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


EINVAL, ENOSYS = -22, -38          # MOY_EINVAL / MOY_ENOSYS of include/moyolo.h
KERNEL_TILED, KERNEL_WREG, KERNEL_DMA, KERNEL_CONV_WS, KERNEL_CONV_S2, KERNEL_CONV_DIRECT = 1, 2, 3, 4, 5, 6     # MOY_KERNEL_*


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().moy_strerror(rc).decode()
        raise MoyoloError(f"{what or 'libmoyolo'} failed: {msg} (code {rc})")

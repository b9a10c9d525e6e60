"""CPU oracle for the step before the path: LetterBox's stretch resize (§8 a1 / §8f rank 3).

TEST INFRASTRUCTURE ONLY -- nothing under mo_yolo_amd/ imports this file.

PARITY UNPINNED against the real dependency: the arithmetic lives in OpenCV (`cv2.resize`, reference
requirements.txt: opencv-python>=4.6.0), which is neither under /root/reference nor installed in the
build container, and the reference holds no resize fixtures.  This file restates OpenCV's published
8-bit INTER_LINEAR algorithm (modules/imgproc/src/resize.cpp, 4.x: `resizeGeneric_` with
HResizeLinear<uchar,int,short,INTER_RESIZE_COEF_SCALE=2048> and the uchar specialisation of
VResizeLinear) and is anchored on the reference call site `LetterBox.__call__`
(ultralytics/data/augment.py:573-576 scaleFill branch + :586-587 `cv2.resize(img, new_unpad,
interpolation=cv2.INTER_LINEAR)`; reached from ultralytics/models/MOTRtrack/predict.py:96-105) and on
known answers: identity, constants, the exact-2x INTER_AREA reroute, and the textbook
[0,255] -> [0,64,191,255] upscaling row.
"""
import numpy as np


def _taps(n_dst, n_src, is_x):
    scale = 1.0 / (np.float64(n_dst) / np.float64(n_src))          # cv::resize: scale = 1 / inv_scale
    d = np.arange(n_dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = f - s.astype(np.float32)
    if is_x:                                                         # x: clamp tap AND zero the weight
        lo = s < 0
        f[lo] = 0; s[lo] = 0
        hi = s >= n_src - 1
        f[hi] = 0; s[hi] = n_src - 1
        s0, s1 = s, np.minimum(s + 1, n_src - 1)
    else:                                                            # y: rows clipped, weights kept
        s0, s1 = np.clip(s, 0, n_src - 1), np.clip(s + 1, 0, n_src - 1)
    a0 = np.clip(np.rint((np.float32(1) - f) * np.float32(2048)), -32768, 32767).astype(np.int64)
    a1 = np.clip(np.rint(f * np.float32(2048)), -32768, 32767).astype(np.int64)
    return s0, s1, a0, a1


def resize_linear_u8(img, out_hw):
    """img uint8 [H, W, C] -> uint8 [Hd, Wd, C] == cv2.resize(img, (Wd, Hd), interpolation=cv2.INTER_LINEAR)."""
    assert img.dtype == np.uint8 and img.ndim == 3
    Hs, Ws, _ = img.shape
    Hd, Wd = out_hw
    if Ws == 2 * Wd and Hs == 2 * Hd:                               # resize.cpp: INTER_LINEAR with iscale 2x2 -> INTER_AREA
        v = img.astype(np.int64)
        return ((v[0::2, 0::2] + v[0::2, 1::2] + v[1::2, 0::2] + v[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    x0, x1, a0, a1 = _taps(Wd, Ws, True)
    y0, y1, b0, b1 = _taps(Hd, Hs, False)
    v = img.astype(np.int64)
    h = v[:, x0] * a0[None, :, None] + v[:, x1] * a1[None, :, None]          # [Hs, Wd, C] int
    r = (((b0[:, None, None] * (h[y0] >> 4)) >> 16) + ((b1[:, None, None] * (h[y1] >> 4)) >> 16) + 2) >> 2
    return np.clip(r, 0, 255).astype(np.uint8)


def letterbox_scalefill(img, new_shape):
    """LetterBox(new_shape, auto=False, scaleFill=True) (data/augment.py:552-597): stretch to new_shape, zero padding
    (dw = dh = 0 so copyMakeBorder adds nothing); a no-op if the size already matches (:586)."""
    if tuple(img.shape[:2]) == tuple(new_shape):
        return img
    return resize_linear_u8(img, new_shape)

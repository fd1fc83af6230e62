"""CPU restatement (numpy, integer arithmetic) of the resize inside the reference's image processor.

TEST INFRASTRUCTURE ONLY (imported by tests/, never by streamvln_amd/).

Reference call site: llava/model/multimodal_encoder/siglip_encoder.py:47-67 -> transformers `resize(..., resample=BICUBIC)` ->
`PIL.Image.resize((384, 384), resample=BICUBIC)`.  The arithmetic lives in a third-party dependency that is not under
/root/reference: **Pillow** (reference pin `pillow==11.2.1`, requirements.txt:97; this container has 12.2.0), file
`src/libImaging/Resample.c` (`precompute_coeffs`, `normalize_coeffs_8bpc`, `ImagingResampleHorizontal_8bpc`,
`ImagingResampleVertical_8bpc`, `bicubic_filter`), unchanged for 8-bit images since Pillow 4.  Its published algorithm, restated:

  * per axis: scale = in/out, filterscale = max(scale, 1), support = 2 * filterscale (bicubic, a = -0.5), ksize = 2*ceil(support)+1;
    for every output index xx: center = (xx + 0.5) * scale, xmin = max(int(center - support + 0.5), 0),
    xmax = min(int(center + support + 0.5), in), weights w_x = bicubic((x + xmin - center + 0.5) / filterscale), normalised by
    their sum (all in double precision);
  * the weights become 22-bit fixed point: k = (int)(w * 2^22 +- 0.5) (round half away from zero);
  * horizontal pass over every input row, then vertical pass over its uint8 result; each output byte is
    clip8((2^21 + sum_x pixel_x * k_x) >> 22) with an arithmetic shift and saturation to [0, 255].

Pinned here against Pillow itself (tests/test_preprocess.py: byte-for-byte on random, structured and odd-sized frames) and against
the reference-generated fixture tests/golden/preprocess.npz (sha256 of whole frames through the reference's SigLipImageProcessor).
"""
from __future__ import annotations

import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def bicubic_filter(x: float) -> float:
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size: int, out_size: int):
    """-> (ksize, xmin [out], count [out], k int32 [out][ksize])  for the full-image box (0, in_size)."""
    in0, in1 = 0.0, float(in_size)
    scale = filterscale = (in1 - in0) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    xmins = np.zeros(out_size, dtype=np.int32)
    counts = np.zeros(out_size, dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = in0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [bicubic_filter((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        xmins[xx], counts[xx] = xmin, xmax
    return ksize, xmins, counts, kk


def _resample_axis0(img: np.ndarray, out_size: int) -> np.ndarray:
    """resample along axis 0 of a uint8 array [n, ...]"""
    n = img.shape[0]
    ksize, xmin, _, kk = precompute_coeffs(n, out_size)
    idx = np.minimum(xmin[:, None] + np.arange(ksize)[None, :], n - 1)          # taps past the count carry weight 0
    acc = np.full((out_size,) + img.shape[1:], 1 << (PRECISION_BITS - 1), dtype=np.int64)
    for t in range(ksize):
        acc += img[idx[:, t]].astype(np.int64) * kk[:, t].reshape((-1,) + (1,) * (img.ndim - 1))
    return np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resize_bicubic_u8(rgb: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """uint8 [H,W,C] -> uint8 [out_h,out_w,C], byte-identical to PIL.Image.resize((out_w, out_h), BICUBIC)."""
    rgb = np.asarray(rgb, dtype=np.uint8)
    if rgb.shape[0] > rgb.shape[1] * 100 and out_h < rgb.shape[0]:      # Pillow 12 `Image.resize`: very tall images shrink vertically first
        rgb = _resample_axis0(rgb, out_h)                                # (absent from the reference's pinned 11.2.1; the HIP path rejects such frames)
    hor = rgb if rgb.shape[1] == out_w else _resample_axis0(rgb.transpose(1, 0, 2), out_w).transpose(1, 0, 2)   # horizontal pass first
    return hor if hor.shape[0] == out_h else _resample_axis0(hor, out_h)


def normalize_lut() -> np.ndarray:
    """fp32 value of every uint8 level after rescale (x * (1/255) in fp64 -> fp32) and normalize ((v - 0.5) / 0.5 in fp32):
    the transformers `rescale` / `normalize` the reference composes (siglip_encoder.py:58-60)."""
    a = (np.arange(256, dtype=np.uint8).astype(np.float64) * (1.0 / 255.0)).astype(np.float32)
    return (a - np.float32(0.5)) / np.float32(0.5)


def siglip_preprocess(rgb: np.ndarray, size: int = 384) -> np.ndarray:
    """uint8 [H,W,3] -> fp32 [3,size,size]: the whole a-1 leg without PIL."""
    u8 = resize_bicubic_u8(rgb, size, size)
    return np.ascontiguousarray(normalize_lut()[u8].transpose(2, 0, 1))

"""TEST INFRASTRUCTURE ONLY -- never imported by the product (magma_amd/).

CPU restatement (numpy, integer arithmetic) of the image-preprocessing step in front of the hot path:
reference magma/transforms.py:121-134 = torchvision ``Resize(n_px, BICUBIC)`` on a PIL image -> ``CenterCrop`` ->
RGB -> ``ToTensor`` -> ``Normalize(CLIP mean/std)``.  ``Resize`` on a PIL image is Pillow's own resampler
(third-party dependency, not vendored in the reference; Pillow 12.2.0 in this image): ImagingResample in
src/libImaging/Resample.c -- ``precompute_coeffs`` (double precision, support = 2 * max(scale, 1), weights
normalised to sum 1), ``normalize_coeffs_8bpc`` (round to 22 fractional bits), horizontal pass then vertical pass,
each ``clip8((1 << 21) + sum(pixel * coeff)) >> 22`` with a uint8 intermediate.

Pinned: tests/test_oracle_pins.py::test_preprocess_matches_pil checks this restatement against PIL itself
(``Image.resize(..., BICUBIC)``) bit for bit on several geometries, so parity for this step is anchored on the
real third-party implementation the reference calls."""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def bicubic_filter(x, a=-0.5):
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size, out_size):
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), dtype=np.int64)
    bounds = np.zeros((out_size, 2), dtype=np.int64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
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
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):   # normalize_coeffs_8bpc
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return kk, bounds


def _pass_rows(img, kk, bounds):
    """resample along axis 1 of an [H, W, C] uint8 array"""
    H, W, C = img.shape
    out = np.empty((H, kk.shape[0], C), dtype=np.uint8)
    src = img.astype(np.int64)
    for xx in range(kk.shape[0]):
        x0, n = bounds[xx]
        acc = (src[:, x0:x0 + n, :] * kk[xx, :n][None, :, None]).sum(1) + (1 << (PRECISION_BITS - 1))
        out[:, xx, :] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return out


def resize_bicubic_u8(img, out_w, out_h):
    """Pillow's Image.resize((out_w, out_h), BICUBIC) on an RGB uint8 array [H, W, 3]."""
    H, W, _ = img.shape
    cur = img
    if out_w != W:
        cur = _pass_rows(cur, *precompute_coeffs(W, out_w))
    if out_h != H:
        cur = _pass_rows(cur.transpose(1, 0, 2), *precompute_coeffs(H, out_h)).transpose(1, 0, 2)
    return np.ascontiguousarray(cur)


def clip_preprocess_u8(img, n_px):
    """[H, W, 3] uint8 -> [3, n_px, n_px] float32, the whole transform of reference transforms.py:121-134."""
    H, W, _ = img.shape
    nw, nh = (n_px, int(n_px * H / W)) if W <= H else (int(n_px * W / H), n_px)
    r = resize_bicubic_u8(img, nw, nh)
    left, top = int(round((nw - n_px) / 2.0)), int(round((nh - n_px) / 2.0))
    c = r[top:top + n_px, left:left + n_px].transpose(2, 0, 1).astype(np.float32) / np.float32(255.0)
    mean = np.asarray(CLIP_MEAN, dtype=np.float32).reshape(3, 1, 1)
    std = np.asarray(CLIP_STD, dtype=np.float32).reshape(3, 1, 1)
    return (c - mean) / std

"""ORACLE — TEST INFRASTRUCTURE ONLY.  Image side of the reference's input pipeline (dataset/base.py:35-44, :55-64):

    train : Compose([Resize(R, BICUBIC), CenterCrop(R), ToTensor(), Normalize(mean, std)])
    eval  : Compose([Resize((R, R), BICUBIC), ToTensor(), Normalize(mean, std)])

applied to a PIL RGB image.  The arithmetic lives in two third-party dependencies that are not vendored in /root/reference
(the reference pins neither): torchvision.transforms (absent from this image; its published algorithm is restated below) and
Pillow's Image.resize (libImaging/Resample.c, present here as Pillow 12.2.0 — tests/golden/make_golden8.py pins this
restatement against it bit for bit).

  * Resize(int): the shorter edge becomes R, the longer int(R * long / short) (torchvision
    transforms/functional.py::_compute_resized_output_size); Resize((R, R)) squashes.  PIL images go to
    `img.resize((w, h), BICUBIC)` — the antialiasing, support-scaled bicubic of ImagingResample, a = -0.5.
  * ImagingResample, 8 bits per channel: per axis, coefficients in double (`precompute_coeffs`), normalised to sum 1,
    rounded to 22-bit fixed point (`normalize_coeffs_8bpc`, PRECISION_BITS = 32 - 8 - 2); horizontal pass first, its uint8
    result feeds the vertical pass; every output = clip8((2^21 + sum pixel * coef) >> 22).
  * CenterCrop(R): top = int(round((h - R) / 2.0)), left likewise — Python's round, i.e. half to even.
  * ToTensor: uint8 -> float32 / 255 (HWC -> CHW);  Normalize: (x - mean) / std in float32.
"""
import math

import numpy as np

MEAN = np.array((0.48145466, 0.4578275, 0.40821073), dtype=np.float32)
STD = np.array((0.26862954, 0.26130258, 0.27577711), dtype=np.float32)
PRECISION_BITS = 32 - 8 - 2


def bicubic_filter(x):
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size, out_size):
    """Resample.c::precompute_coeffs with box (0, in_size) + normalize_coeffs_8bpc -> (ksize, bounds [out,2], int coeffs [out,ksize])."""
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
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
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


def _pass(img, out_size, axis):
    """One ImagingResample pass along `axis` (0 = vertical, 1 = horizontal) of a uint8 [H, W, C] image."""
    in_size = img.shape[axis]
    _, bounds, kk = precompute_coeffs(in_size, out_size)
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((out_size,) + src.shape[1:], dtype=np.uint8)
    for xx in range(out_size):
        xmin, n = bounds[xx]
        acc = np.tensordot(kk[xx, :n].astype(np.int64), src[xmin:xmin + n], axes=(0, 0)) + (1 << (PRECISION_BITS - 1))
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def pil_resize_bicubic(img, out_w, out_h):
    """Image.resize((out_w, out_h), BICUBIC) of a uint8 RGB array [H, W, 3]: horizontal pass, then vertical (each skipped when
    the size does not change, as ImagingResample does)."""
    h, w = img.shape[:2]
    if out_w != w:
        img = _pass(img, out_w, 1)
    if out_h != h:
        img = _pass(img, out_h, 0)
    return img


def resized_size(h, w, R, train):
    """-> (new_h, new_w) of the Resize step."""
    if not train:
        return R, R
    short, long_ = (w, h) if w <= h else (h, w)
    new_short, new_long = R, int(R * long_ / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)


def crop_origin(h, w, R):
    return int(round((h - R) / 2.0)), int(round((w - R) / 2.0))


def transform_u8(img, R, train):
    """The uint8 [R, R, 3] image that reaches ToTensor."""
    h, w = img.shape[:2]
    nh, nw = resized_size(h, w, R, train)
    out = pil_resize_bicubic(img, nw, nh)
    if train:
        top, left = crop_origin(nh, nw, R)
        out = out[top:top + R, left:left + R]
    return out


def transform(img, R=224, train=True):
    """uint8 RGB [H, W, 3] -> float32 [3, R, R], the tensor BaseDataset._load_image returns (dataset/base.py:55-64)."""
    u8 = transform_u8(img, R, train)
    x = u8.astype(np.float32) / np.float32(255)
    x = (x - MEAN) / STD
    return np.ascontiguousarray(x.transpose(2, 0, 1))

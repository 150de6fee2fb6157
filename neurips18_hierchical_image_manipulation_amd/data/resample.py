"""Index and coefficient tables of the two Pillow resamplers the reference's loader uses (host side, tiny).

The reference resizes label / instance maps with ``Image.NEAREST`` and photographs with ``Image.BICUBIC``
(data/base_dataset.py:243-268, 373-377 -> ``PIL.Image.resize``; Pillow is the third-party dependency, pinned here at
12.2.0).  The pixel work runs on the GPU (csrc/him_data.hip); what each output row / column reads -- a source index
for NEAREST, a window of 22-bit fixed-point weights for the antialiased BICUBIC -- is a few hundred numbers per image
and is computed here, in double precision and in the operation order of Pillow's published algorithm
(src/libImaging/Geometry.c ``ImagingScaleAffine``; src/libImaging/Resample.c ``precompute_coeffs`` /
``normalize_coeffs_8bpc``), so that the device result is bit-identical to Pillow's (tests compare against Pillow).
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2     # Resample.c: 8 bits of pixel, 2 bits of head-room for the negative bicubic lobes
BICUBIC_SUPPORT = 2.0


_NEAREST_CACHE = {}


def nearest_table(in_size, out_size, sixteen_bit=False):
    """Source index of every output position for a NEAREST resize of ``in_size`` -> ``out_size`` samples.

    For 8- and 32-bit images Pillow walks ``xo = a/2, a/2 + a, ...`` by repeated addition (a = in/out) and truncates
    (``ImagingScaleAffine``); the running sum is reproduced as is -- a closed form differs in the last bit for some
    sizes.  16-bit images (mode ``I;16``, the Cityscapes instance maps) are a "special" storage type and take
    Pillow's generic transform instead, which evaluates ``a * (i + 0.5)`` per pixel (``affine_transform``)."""
    key = (int(in_size), int(out_size), bool(sixteen_bit))
    hit = _NEAREST_CACHE.get(key)
    if hit is not None:
        return hit
    step = float(in_size) / out_size
    tab = np.empty(out_size, np.int32)
    pos = 0.0 + step * 0.5
    for i in range(out_size):
        if sixteen_bit:
            pos = step * (i + 0.5) + 0.0 * 0.5 + 0.0
        src = -1 if pos < 0.0 else int(pos)
        tab[i] = min(max(src, 0), in_size - 1)   # positions outside the image keep Pillow's fill; none occur for a full-box resize
        if not sixteen_bit:
            pos += step
    if len(_NEAREST_CACHE) > 2048:
        _NEAREST_CACHE.clear()
    _NEAREST_CACHE[key] = tab
    return tab


def _bicubic(x):
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def bicubic_tables_scalar(in_size, out_size):
    """Straight transcription of Pillow's loop (the cross-check of the vectorised ``bicubic_tables`` below)."""
    scale = float(in_size) / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = BICUBIC_SUPPORT * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    first = np.zeros(out_size, np.int32)
    count = np.zeros(out_size, np.int32)
    weights = np.zeros((out_size, ksize), np.int32)
    inv = 1.0 / filterscale
    one = float(1 << PRECISION_BITS)
    for i in range(out_size):
        center = 0.0 + (i + 0.5) * scale
        lo = int(center - support + 0.5)
        if lo < 0:
            lo = 0
        hi = int(center + support + 0.5)
        if hi > in_size:
            hi = in_size
        n = hi - lo
        w = [_bicubic((k + lo - center + 0.5) * inv) for k in range(n)]
        total = 0.0
        for v in w:
            total += v
        for k in range(n):
            v = w[k] / total if total != 0.0 else w[k]
            weights[i, k] = int(-0.5 + v * one) if v < 0 else int(0.5 + v * one)
        first[i], count[i] = lo, n
    return first, count, weights, ksize


_TABLE_CACHE = {}


def bicubic_tables(in_size, out_size):
    """(first [out], count [out], weights [out, ksize] int32, ksize) of the antialiased bicubic resample.

    Output sample i = clip8((2^21 + sum_k weights[i, k] * src[first[i] + k]) >> 22).  Same doubles in the same order as
    Pillow's loop, evaluated for all output samples at once (the running sum of a row is numpy's sequential cumsum)."""
    key = (int(in_size), int(out_size))
    hit = _TABLE_CACHE.get(key)
    if hit is not None:
        return hit
    scale = float(in_size) / out_size
    filterscale = scale if scale >= 1.0 else 1.0
    support = BICUBIC_SUPPORT * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    inv = 1.0 / filterscale
    one = float(1 << PRECISION_BITS)
    center = 0.0 + (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    lo = np.maximum(np.trunc(center - support + 0.5).astype(np.int64), 0)
    hi = np.minimum(np.trunc(center + support + 0.5).astype(np.int64), in_size)
    n = hi - lo
    k = np.arange(ksize, dtype=np.int64)[None, :]
    x = np.abs(((k + lo[:, None]).astype(np.float64) - center[:, None] + 0.5) * inv)
    a = -0.5
    near = ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    far = (((x - 5) * x + 8) * x - 4) * a
    w = np.where(x < 1.0, near, np.where(x < 2.0, far, 0.0))
    w = np.where(k < n[:, None], w, 0.0)
    total = np.cumsum(w, axis=1)[:, -1:]
    v = np.where(total != 0.0, w / np.where(total != 0.0, total, 1.0), w)
    weights = np.where(v < 0, np.trunc(-0.5 + v * one), np.trunc(0.5 + v * one)).astype(np.int32)
    weights = np.where(k < n[:, None], weights, 0).astype(np.int32)
    out = (lo.astype(np.int32), n.astype(np.int32), weights, ksize)
    if len(_TABLE_CACHE) > 512:
        _TABLE_CACHE.clear()
    _TABLE_CACHE[key] = out
    return out


def pil_crop_box(box):
    """``Image.crop`` rounds a float box half-to-even (``map(int, map(round, box))``, Image.py ``_crop``)."""
    return tuple(int(round(v)) for v in box)

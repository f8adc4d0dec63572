"""Fixed-point LANCZOS coefficient tables of Pillow's 8-bit `Image.resize`.

The reference's PILRenderer draws on an `anti_aliasing`-times larger canvas and
shrinks it with `Image.resize(..., ANTIALIAS)` (reference:
spriteworld/renderers/pil_renderer.py:50-51,84).  The arithmetic is Pillow's
(libImaging/Resample.c `precompute_coeffs`, `normalize_coeffs_8bpc`; pinned for
Pillow 12.2.0 in SURVEY.md A.6): per output index a window (xmin, count) of
source pixels and `count` weights lanczos3((x + xmin - center + 0.5) / scale),
normalised by their float64 running sum and rounded to 22-bit fixed point.  The
GPU consumes these tables verbatim (`swb_upload_resample`), so the integer
resample on device is exact.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def _sinc(x):
  if x == 0.0:
    return 1.0
  x = x * math.pi
  return math.sin(x) / x


def _lanczos3(x):
  if -3.0 <= x < 3.0:
    return _sinc(x) * _sinc(x / 3)
  return 0.0


def resample_tables(in_size, out_size):
  """Returns (bounds i32[out, 2] = (xmin, count), coeffs i32[out, ksize])."""
  scale = float(np.float32(in_size) - np.float32(0.0)) / out_size
  filterscale = max(scale, 1.0)
  support = 3.0 * filterscale
  ksize = int(math.ceil(support)) * 2 + 1
  bounds = np.zeros((out_size, 2), dtype=np.int32)
  coeffs = np.zeros((out_size, ksize), dtype=np.int32)
  inv = 1.0 / filterscale
  for o in range(out_size):
    center = 0.0 + (o + 0.5) * scale
    xmin = max(int(center - support + 0.5), 0)
    count = min(int(center + support + 0.5), in_size) - xmin
    weights = [_lanczos3((x + xmin - center + 0.5) * inv) for x in range(count)]
    total = 0.0
    for w in weights:
      total += w
    for x, w in enumerate(weights):
      if total != 0.0:
        w = w / total
      coeffs[o, x] = int(-0.5 + w * (1 << PRECISION_BITS)) if w < 0 else int(
          0.5 + w * (1 << PRECISION_BITS))
    bounds[o] = (xmin, count)
  return bounds, coeffs

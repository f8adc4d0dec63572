"""Synthetic episode pools with the factor distributions of the shipped configs.

No reference code or data is needed: sprites are drawn with numpy in the value
ranges and dtypes the COBRA configs use (reference:
spriteworld/configs/cobra/clustering.py:41-46,71-78,
spriteworld/configs/cobra/goal_finding_more_distractors.py:54-74,
spriteworld/configs/examples/goal_finding_embodied.py:82-93): positions and
colours are np.float32 samples of U[lo, hi), shapes are drawn from a list, and
the colour is mapped with the renderer's hsv_to_rgb.  Used by bench.py, smoke()
and the GPU parity tests (both the oracle and the HIP engine consume the same
pool arrays).
"""
import math

import numpy as np

from spriteworld_amd import lowering
from spriteworld_amd import renderers
from spriteworld_amd import shapes as _shapes


def _u32(rng, lo, hi):
  return np.float32(rng.uniform(lo, hi))


def make_pool(rng, n_entries, sprites_per_entry, hue_ranges, labels, n_tasks=1,
              shape_names=('square', 'triangle', 'circle'), scales=(0.13,), angles=(0,),
              xy_range=(0.1, 0.9), body=None, shuffle=True):
  """Pool of `n_entries` episodes.

  hue_ranges: per sprite slot (lo, hi) of c0; labels: per sprite slot, per task, the int8 label
  (FindGoal membership 0/1 or cluster id / -1).  `body`: optional dict(scale=, hue=(lo,hi),
  shape='circle') appended as the last (foreground) sprite, as the Embodied configs do.
  """
  S = sprites_per_entry + (1 if body else 0)
  pool = lowering.Pool(n_entries, S, n_tasks)
  pool.n_sprites[:] = S
  shape_ids = np.array([_shapes.shape_index(s) for s in shape_names], np.int32)
  for e in range(n_entries):
    order = rng.permutation(sprites_per_entry) if shuffle else np.arange(sprites_per_entry)
    for slot, src in enumerate(order):
      lo, hi = hue_ranges[src]
      hsv = (_u32(rng, lo, hi), _u32(rng, 0.3, 1.0), _u32(rng, 0.9, 1.0))
      _fill(pool, e, slot, rng, xy_range, shape_ids[rng.integers(len(shape_ids))],
            scales[rng.integers(len(scales))], angles[rng.integers(len(angles))], hsv)
      for t in range(n_tasks):
        pool.label[e, t, slot] = labels[src][t]
    if body:
      lo, hi = body.get('hue', (0.0, 1.0))
      hsv = (_u32(rng, lo, hi), np.float32(1.0), np.float32(1.0))
      _fill(pool, e, S - 1, rng, xy_range, _shapes.shape_index(body.get('shape', 'circle')),
            body.get('scale', 0.07), 0, hsv)
      for t in range(n_tasks):
        pool.label[e, t, S - 1] = body.get('label', 0)
  return pool


def _fill(pool, e, s, rng, xy_range, shape_id, scale, angle, hsv):
  pool.x[e, s] = float(_u32(rng, xy_range[0], xy_range[1]))
  pool.y[e, s] = float(_u32(rng, xy_range[0], xy_range[1]))
  pool.shape[e, s] = shape_id
  pool.scale[e, s] = float(scale)
  th = math.radians(angle)
  pool.cos_a[e, s], pool.sin_a[e, s] = math.cos(th), math.sin(th)
  pool.angle[e, s] = float(angle)
  pool.rgb[e, s, :3] = np.asarray(renderers.hsv_to_rgb(hsv), dtype=np.uint8)
  pool.color[e, s] = [float(c) for c in hsv]

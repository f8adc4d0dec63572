"""Pure-Python restatement of the device reset sampler (spriteworld_amd/csrc/swb_sampler.hip.inc).

Test infrastructure: draws the same Philox4x32-10 stream in the documented order and builds the
sprites with the HOST classes (Sprite, the renderer's color_to_rgb, the task's filters), so a
bit-for-bit comparison with swb_get_pool checks the RNG, the draw order, the value dtypes, the
hsv colour map and the labels at once.
"""
import math

import numpy as np

from spriteworld_amd import _abi

M32 = 0xFFFFFFFF


def philox4x32_10(ctr, key):
  c0, c1, c2, c3 = ctr
  k0, k1 = key
  for _ in range(10):
    p0 = 0xD2511F53 * c0
    p1 = 0xCD9E8D57 * c2
    c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & M32, p1 & M32, ((p0 >> 32) ^ c3 ^ k1) & M32, p0 & M32
    k0 = (k0 + 0x9E3779B9) & M32
    k1 = (k1 + 0xBB67AE85) & M32
  return c0, c1, c2, c3


class Stream(object):

  def __init__(self, seed, entry):
    self.key = (seed & M32, (seed >> 32) & M32)
    self.entry, self.entry_hi, self.block, self.buf = entry & M32, entry >> 32, 0, []

  def u32(self):
    if not self.buf:
      self.buf = list(philox4x32_10((self.entry, self.block, self.entry_hi, 0), self.key))
      self.block += 1
    return self.buf.pop(0)

  def uniform(self):
    a, b = self.u32() >> 5, self.u32() >> 6
    return (a * 67108864.0 + b) / 9007199254740992.0


def _draw(rng, fac, np_dtype=None):
  if fac.kind == _abi.FACTOR_DISCRETE:
    return fac.cand[rng.u32() % fac.n]          # Python float
  val = fac.lo + (fac.hi - fac.lo) * rng.uniform()
  if fac.kind == _abi.FACTOR_UNIFORM_INT:
    return int(np.asarray(val).astype(np.int64))
  return np.float32(val)


def sample_pool(spec, n_entries, max_sprites, seed, to_rgb, label_fns, shape_names, first_entry=0):
  """Returns a dict of arrays laid out like lowering.Pool."""
  P, S, T = n_entries, max_sprites, len(label_fns)
  out = dict(n_sprites=np.zeros(P, np.int32), x=np.zeros((P, S)), y=np.zeros((P, S)), x_vel=np.zeros((P, S)),
             y_vel=np.zeros((P, S)), scale=np.ones((P, S)), cos_a=np.ones((P, S)), sin_a=np.zeros((P, S)),
             angle=np.zeros((P, S)), shape=np.zeros((P, S), np.int32), rgb=np.zeros((P, S, 4), np.uint8),
             color=np.zeros((P, S, 3)), label=np.zeros((P, T, S), np.int8))
  for e in range(P):
    rng = Stream(seed, first_entry + e)
    if spec.n_alternatives > 0:
      alt = spec.alternatives[rng.u32() % spec.n_alternatives if spec.n_alternatives > 1 else 0]
      order = [alt.group[g] for g in range(alt.n)]
    else:
      order = list(range(spec.n_groups))
    counts, n = [], 0
    for g in order:
      grp = spec.groups[g]
      c = grp.count_min + rng.u32() % (grp.count_max - grp.count_min + 1)
      c = min(c, S - n)
      counts.append(c)
      n += c
    slot = list(range(16))
    m = sum(counts[:spec.shuffle])
    if m > 1:
      for i in range(m - 1, 0, -1):
        j = rng.u32() % (i + 1)
        slot[i], slot[j] = slot[j], slot[i]
    out['n_sprites'][e] = n
    k = 0
    for gi, g in enumerate(order):
      grp = spec.groups[g]
      for _ in range(counts[gi]):
        s = slot[k]
        k += 1
        fv = [None] * _abi.SWB_N_FACTORS
        fv[0], fv[1] = _draw(rng, grp.factors[0]), _draw(rng, grp.factors[1])
        shape = grp.shapes[rng.u32() % grp.n_shapes]
        for i in range(2, _abi.SWB_N_FACTORS):
          fv[i] = _draw(rng, grp.factors[i])
        for h in range(grp.n_holdouts):
          ho = grp.holdouts[h]
          for _ in range(9999):
            if not all(ho.lo[i] <= fv[i] < ho.hi[i] for i in range(_abi.SWB_N_FACTORS) if (ho.box_mask >> i) & 1):
              break
            for i in range(_abi.SWB_N_FACTORS):
              if (ho.redraw_mask >> i) & 1:
                fv[i] = _draw(rng, grp.factors[i])
        x, y, scale, angle, c0, c1, c2, xv, yv = fv
        out['x'][e, s], out['y'][e, s] = float(x), float(y)
        out['x_vel'][e, s], out['y_vel'][e, s] = float(xv), float(yv)
        out['shape'][e, s], out['scale'][e, s], out['angle'][e, s] = shape, float(scale), float(angle)
        th = math.radians(angle)
        out['cos_a'][e, s], out['sin_a'][e, s] = math.cos(th), math.sin(th)
        out['color'][e, s] = [float(c0), float(c1), float(c2)]
        out['rgb'][e, s, :3] = np.asarray(to_rgb((c0, c1, c2))).astype(np.uint8)
        factors = dict(x=x, y=y, shape=shape_names[shape], angle=angle, scale=scale, c0=c0, c1=c1, c2=c2,
                       x_vel=xv, y_vel=yv)
        for t, fn in enumerate(label_fns):
          out['label'][e, t, s] = fn(factors)
  return out

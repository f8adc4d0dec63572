"""Declarative sprite generator that the engine can sample on the device (swb_sample_pool).

The reference's `init_sprites` is an opaque Python callable (environment.py:68,75), so resets cost
host time per episode.  Every shipped config builds it from
`generate_sprites(Product([Continuous/Discrete ...]), num_sprites)` groups that are chained and
optionally shuffled (sprite_generators.py:27-70,101-128).  `DeviceSampler` takes those same
ingredients, keeps them inspectable, and lowers them to `swb_sampler` (include/swb.h):

    sampler = DeviceSampler([(target_factors, 2), (distractor_factors, (1, 4))], shuffle=True)
    env = BatchedEnvironment(init_sprites=sampler, ...)      # pool drawn by a HIP kernel
    env.refill_pool()                                        # fresh episodes, no host work

Calling the object samples on the host with numpy (same marginals), so it is also a valid
`init_sprites` for the reference's own Environment.  Device draws come from Philox4x32-10 streams:
the distribution is the reference's, the bit stream is not MT19937's.
"""
import math

import numpy as np

from spriteworld_amd import _abi
from spriteworld_amd import shapes as _shapes
from spriteworld_amd import sprite as sprite_lib
from spriteworld_amd.lowering import LoweringError, subtasks_of, _label_of

_FACTOR_KEYS = _abi.FACTOR_ORDER
_DEFAULTS = dict(x=0.5, y=0.5, shape='square', angle=0, scale=0.1, c0=0, c1=0, c2=0, x_vel=0.0, y_vel=0.0)


def _lower_factor(dst, key, leaf):
  """Continuous / Discrete leaf (or None: the Sprite default) -> swb_factor."""
  if leaf is None:
    dst.kind, dst.n = _abi.FACTOR_DISCRETE, 1
    dst.cand[0] = float(_DEFAULTS[key])
    return [_DEFAULTS[key]]
  if type(leaf).__name__ == 'Continuous':
    dt = np.dtype(leaf.dtype)
    if dt == np.float32:
      dst.kind = _abi.FACTOR_UNIFORM_F32
    elif dt.kind in 'iu':
      dst.kind = _abi.FACTOR_UNIFORM_INT
    else:
      raise LoweringError('factor %s: dtype %s is not sampled on the device' % (key, dt))
    dst.lo, dst.hi = float(leaf.minval), float(leaf.maxval)
    return None
  if getattr(leaf, 'probs', None) is not None:
    raise LoweringError('factor %s: weighted Discrete factors are not sampled on the device' % key)
  cands = list(leaf.candidates)
  if not 1 <= len(cands) <= _abi.SWB_MAX_CANDIDATES:
    raise LoweringError('factor %s: 1..%d candidates' % (key, _abi.SWB_MAX_CANDIDATES))
  if not all(type(c) in (float, int) for c in cands):
    raise LoweringError('factor %s: Discrete candidates must be Python floats or ints' % key)
  dst.kind, dst.n = _abi.FACTOR_DISCRETE, len(cands)
  for i, c in enumerate(cands):
    dst.cand[i] = float(c)
  return cands


def _box_of(dist):
  """A hold-out region as {key: (lo, hi)}: a Continuous leaf or a Product of them."""
  name = type(dist).__name__
  if name == 'Continuous':
    return {dist.key: (float(dist.minval), float(dist.maxval))}
  if name == 'Product':
    out = {}
    for c in dist.components:
      out.update(_box_of(c))
    return out
  raise LoweringError('hold-out regions must be Continuous ranges (got %s)' % name)


def _marginals(dist, holdouts=None):
  """Flattens a Product / SetMinus tree of Continuous/Discrete leaves into {key: leaf}.

  SetMinus nodes append (redrawn keys, {key: (lo, hi)}) to `holdouts`."""
  name = type(dist).__name__
  if name == 'Product':
    out = {}
    for c in dist.components:
      sub = _marginals(c, holdouts)
      if set(sub) & set(out):
        raise LoweringError('factor %s appears twice' % sorted(set(sub) & set(out)))
      out.update(sub)
    return out
  if name in ('Continuous', 'Discrete'):
    return {dist.key: dist}
  if name == 'SetMinus' and holdouts is not None:
    inner = []
    out = _marginals(dist.base, inner)
    if inner:
      raise LoweringError('nested SetMinus distributions are not sampled on the device')
    holdouts.append((sorted(out), _box_of(dist.hold_out)))
    return out
  raise LoweringError('%s cannot be sampled on the device (Product / SetMinus / Continuous / Discrete)' % name)


class DeviceSampler(object):
  """`groups`: list of (factor_distribution, num_sprites); num_sprites is an int or a (lo, hi)
  pair meaning np.random.randint(lo, hi).  `shuffle`: sprite_generators.shuffle of the z-order --
  True: all sprites; an int k: only the sprites of the first k groups (the rest keep their place
  on top, like the agent body of examples/goal_finding_embodied.py:88-93).  `alternatives`:
  sprite_generators.sample_generator -- a list of lists of group indices; each episode generates
  the groups of one list chosen uniformly (default: all groups, in order)."""

  def __init__(self, groups, shuffle=False, seed=0, alternatives=None):
    self.groups, self._holdouts = [], []
    for dist, count in groups:
      if isinstance(count, (tuple, list)):
        lo, hi = int(count[0]), int(count[1]) - 1
      else:
        lo = hi = int(count)
      if lo < 0 or hi < lo:
        raise ValueError('bad sprite count %r' % (count,))
      holdouts = []
      marg = _marginals(dist, holdouts)
      if len(holdouts) > _abi.SWB_MAX_HOLDOUTS:
        raise LoweringError('at most %d SetMinus nodes per sprite group' % _abi.SWB_MAX_HOLDOUTS)
      self.groups.append((dist, lo, hi, marg))
      self._holdouts.append(holdouts)
    if not 1 <= len(self.groups) <= _abi.SWB_MAX_GROUPS:
      raise LoweringError('1..%d sprite groups are supported' % _abi.SWB_MAX_GROUPS)
    self.shuffle = _abi.SWB_MAX_GROUPS if shuffle is True else int(shuffle)
    if self.shuffle < 0:
      raise ValueError('shuffle must be a bool or a number of leading groups')
    self.alternatives = None if alternatives is None else [list(map(int, a)) for a in alternatives]
    if self.alternatives is not None:
      if not 1 <= len(self.alternatives) <= _abi.SWB_MAX_ALTERNATIVES:
        raise LoweringError('1..%d alternatives are supported' % _abi.SWB_MAX_ALTERNATIVES)
      for a in self.alternatives:
        if not 1 <= len(a) <= _abi.SWB_MAX_GROUPS or not all(0 <= g < len(self.groups) for g in a):
          raise ValueError('bad alternative %r' % (a,))
    self.seed = int(seed)
    self._draws = 0

  def _lists(self):
    return self.alternatives if self.alternatives is not None else [list(range(len(self.groups)))]

  @property
  def max_sprites(self):
    return max(sum(self.groups[g][2] for g in lst) for lst in self._lists())

  def next_seed(self):
    """A fresh 64-bit Philox key per pool (splitmix64 of seed and draw counter)."""
    z = (self.seed * 0x9E3779B97F4A7C15 + self._draws + 1) & 0xFFFFFFFFFFFFFFFF
    self._draws += 1
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return z ^ (z >> 31)

  # ---------------------------------------------------------------- host sampling (numpy)
  def __call__(self):
    lists = self._lists()
    chosen = lists[np.random.randint(len(lists))] if len(lists) > 1 else lists[0]
    sprites, n_shuffled = [], 0
    for g, gi in enumerate(chosen):
      dist, lo, hi, _ = self.groups[gi]
      n = lo if lo == hi else np.random.randint(lo, hi + 1)
      sprites.extend(sprite_lib.Sprite(**dist.sample()) for _ in range(n))
      if g < self.shuffle:
        n_shuffled = len(sprites)
    if n_shuffled > 1:
      order = np.random.permutation(n_shuffled)
      sprites = [sprites[i] for i in order] + sprites[n_shuffled:]
    return sprites

  # ---------------------------------------------------------------- lowering
  def _group_label(self, sub, dist, marg):
    """Task label of a group's sprites; must not depend on the sampled values."""
    rng_state = np.random.get_state()      # the probes must not advance the caller's global numpy stream
    try:
      return self._group_label_probed(sub, dist, marg)
    finally:
      np.random.set_state(rng_state)

  def _group_label_probed(self, sub, dist, marg):
    probes = [dist.sample() for _ in range(256)]
    for key, leaf in marg.items():  # range ends of every continuous factor
      if type(leaf).__name__ == 'Continuous' and leaf.maxval > leaf.minval:
        if np.dtype(leaf.dtype) == np.float32:
          lo = np.float32(leaf.minval)
          hi = np.nextafter(np.float32(leaf.maxval), np.float32(-np.inf))
        else:
          lo = np.asarray(leaf.minval).astype(leaf.dtype)
          hi = np.asarray(np.nextafter(float(leaf.maxval), -np.inf)).astype(leaf.dtype)
        for v in (lo, hi):
          f = dist.sample()
          f[key] = v
          probes.append(f)
    labels = set()
    for f in probes:
      full = dict(_DEFAULTS)
      full.update(f)
      labels.add(_label_of(sub, sprite_lib.Sprite(**full)))
    if len(labels) != 1:
      raise LoweringError('task label of a sprite group depends on the sampled factors (%s): sample this '
                          'generator on the host instead' % sorted(labels))
    return labels.pop()

  def scale_kinds(self):
    """swb_factor kind of every group's `scale` factor (FACTOR_UNIFORM_F32: the sprites carry np.float32 scales)."""
    kinds = []
    for _, _, _, marg in self.groups:
      fac = _abi.SwbFactor()
      _lower_factor(fac, 'scale', marg.get('scale'))
      kinds.append(fac.kind)
    return kinds

  def lower(self, task, renderers):
    from spriteworld_amd import lowering
    spec = _abi.SwbSampler()
    spec.n_groups = len(self.groups)
    spec.shuffle = self.shuffle
    spec.n_alternatives = 0 if self.alternatives is None else len(self.alternatives)
    for i, lst in enumerate(self.alternatives or []):
      spec.alternatives[i].n = len(lst)
      for j, g in enumerate(lst):
        spec.alternatives[i].group[j] = g
    _, pil = lowering.find_pil_renderer(renderers)
    to_rgb = getattr(pil, '_color_to_rgb', None) if pil is not None else None
    probe = (0.3, 0.6, 0.9)
    if to_rgb is None or tuple(to_rgb(probe)) == probe:
      spec.color_map = 0
    elif getattr(to_rgb, '__name__', '') == 'hsv_to_rgb':
      spec.color_map = 1
    else:
      raise LoweringError('color_to_rgb %r is not available on the device (identity or hsv_to_rgb)' % (to_rgb,))
    for d in range(360):  # sprite.py:147-151 rotates by radians of the stored degrees
      spec.deg_cos[d], spec.deg_sin[d] = math.cos(math.radians(d)), math.sin(math.radians(d))
    subs = subtasks_of(task)
    for sub in subs:      # labels that depend on WHERE a sprite is drawn (and then moved) are tabulated per episode on the host
      xc, yc = lowering.position_cuts(sub, True)
      if xc or yc:
        raise LoweringError('a task filter keys on position: its episodes are sampled on the host (lowering.lower_episodes '
                            'tabulates each sprite\'s label over the filter\'s position grid)')
    for g, (dist, lo, hi, marg) in enumerate(self.groups):
      grp = spec.groups[g]
      grp.count_min, grp.count_max = lo, hi
      unknown = set(marg) - set(_FACTOR_KEYS) - {'shape'}
      if unknown:
        raise LoweringError('unknown sprite factors %s' % sorted(unknown))
      for key in _FACTOR_KEYS:
        fac = grp.factor(key)
        cands = _lower_factor(fac, key, marg.get(key))
        if key in ('x', 'y') and fac.kind != _abi.FACTOR_UNIFORM_F32:
          raise LoweringError('%s must be a float32 Continuous factor' % key)
        if key == 'angle':
          if fac.kind == _abi.FACTOR_UNIFORM_F32 or (
              fac.kind == _abi.FACTOR_UNIFORM_INT and not 0 <= fac.lo <= fac.hi <= 360):
            raise LoweringError('angle must be Discrete or integer degrees within [0, 360]')
          for i, c in enumerate(cands or []):
            grp.cos_a[i], grp.sin_a[i] = math.cos(math.radians(c)), math.sin(math.radians(c))
        if spec.color_map == 1 and key in ('c0', 'c1', 'c2') and fac.kind == _abi.FACTOR_UNIFORM_INT:
          raise LoweringError('hsv colours must be float factors')
      grp.n_holdouts = len(self._holdouts[g])
      for i, (redrawn, box) in enumerate(self._holdouts[g]):
        ho = grp.holdouts[i]
        if 'shape' in box:
          raise LoweringError('hold-out regions over shape are not sampled on the device')
        ho.redraw_mask = sum(1 << _abi.FACTOR_ORDER.index(k) for k in redrawn if k != 'shape')
        ho.box_mask = sum(1 << _abi.FACTOR_ORDER.index(k) for k in box)
        for k, (blo, bhi) in box.items():
          ho.lo[_abi.FACTOR_ORDER.index(k)], ho.hi[_abi.FACTOR_ORDER.index(k)] = blo, bhi
        if 'shape' in redrawn:
          raise LoweringError('SetMinus over a base that includes shape is not sampled on the device')
      leaf = marg.get('shape')
      if leaf is None:
        names = [_DEFAULTS['shape']]
      elif type(leaf).__name__ == 'Discrete' and getattr(leaf, 'probs', None) is None:
        names = list(leaf.candidates)
      else:
        raise LoweringError('shape must be an unweighted Discrete factor')
      if not 1 <= len(names) <= _abi.SWB_MAX_CANDIDATES:
        raise LoweringError('shape: 1..%d candidates' % _abi.SWB_MAX_CANDIDATES)
      grp.n_shapes = len(names)
      for i, name in enumerate(names):
        grp.shapes[i] = _shapes.shape_index(name)
      for t, sub in enumerate(subs):
        grp.label[t] = self._group_label(sub, dist, marg)
    return spec


# ------------------------------------------------------------------------------ closures -> sampler
def _closure(fn):
  return dict(zip(fn.__code__.co_freevars, (c.cell_contents for c in fn.__closure__ or ())))


def _count_of(num_sprites):
  """int, or the (lo, hi) of a `lambda: np.random.randint(lo, hi)` (cobra/exploration.py:58)."""
  if not callable(num_sprites):
    return int(num_sprites)
  code = getattr(num_sprites, '__code__', None)
  if code is not None and code.co_argcount == 0 and code.co_names[-1:] == ('randint',):
    ints = [c for c in code.co_consts if isinstance(c, int) and not isinstance(c, bool)]
    if len(ints) == 2 and not num_sprites.__closure__:
      return (ints[0], ints[1])
  raise LoweringError('num_sprites callable is not a plain `lambda: np.random.randint(lo, hi)`')


def _flatten(fn):
  """Generator closure -> (groups, n_leading_groups_shuffled or None, alternatives or None)."""
  kind = getattr(fn, '__qualname__', '').split('.')[0]
  cells = _closure(fn) if getattr(fn, '__code__', None) is not None else {}
  if kind == 'generate_sprites':
    return [(cells['factor_dist'], _count_of(cells['num_sprites']))], None, None
  if kind == 'shuffle':
    groups, inner, alts = _flatten(cells.get('sprite_generator', cells.get('generator')))
    if inner is not None:
      raise LoweringError('shuffle of an already shuffled generator is not sampled on the device')
    return groups, _abi.SWB_MAX_GROUPS, alts
  if kind == 'chain_generators':
    gens = cells.get('sprite_generators', cells.get('generators'))
    groups, shuffled = [], None
    for i, g in enumerate(gens):
      sub, sh, alts = _flatten(g)
      if alts is not None:
        raise LoweringError('sample_generator inside chain_generators is not sampled on the device')
      if sh is not None:
        if i != 0:
          raise LoweringError('only a leading shuffled sub-generator is sampled on the device')
        shuffled = len(sub)
      groups.extend(sub)
    return groups, shuffled, None
  if kind == 'sample_generator':
    if cells.get('p') is not None:
      raise LoweringError('weighted sample_generator is not sampled on the device')
    groups, alts, seen = [], [], {}
    for g in cells.get('sprite_generators', cells.get('generators')):
      sub, sh, inner = _flatten(g)
      if sh is not None or inner is not None:
        raise LoweringError('nested shuffle / sample_generator is not sampled on the device')
      lst = []
      for dist, count in sub:                  # the same (distribution, count) is one group
        key = (id(dist), count)
        if key not in seen:
          seen[key] = len(groups)
          groups.append((dist, count))
        lst.append(seen[key])
      alts.append(lst)
    return groups, None, alts
  raise LoweringError('%r is not built from generate_sprites / chain_generators / sample_generator / shuffle' % (fn,))


def from_generator(init_sprites, seed=0):
  """DeviceSampler equivalent to a generator built with the reference's (or this package's)
  `sprite_generators.generate_sprites / chain_generators / sample_generator / shuffle`, by reading
  their closures.  Raises LoweringError for anything else (hand-written callables, weighted
  choices, ...)."""
  if isinstance(init_sprites, DeviceSampler):
    return init_sprites
  groups, shuffled, alts = _flatten(init_sprites)
  return DeviceSampler(groups, shuffle=shuffled or 0, seed=seed, alternatives=alts)

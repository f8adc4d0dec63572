"""Distributions over sprite factors (host side, reset time only).

Small mirrors of the reference's distribution algebra (reference:
spriteworld/factor_distributions.py:81-120 Continuous, :123-158 Discrete,
:161-209 Mixture, :212-265 Intersection, :268-310 Product, :313-358 SetMinus):
`sample(rng=None)` returns {factor: value}, `contains(spec)` tests membership,
`keys` is the set of factor names.  They run on the host when an episode pool is
drawn; tasks use `contains` for the per-sprite filter / cluster labels that
`lowering.lower_episodes` bakes into the pool.  The reference's own classes can
be used instead -- lowering is duck-typed.
"""
import numpy as np

_MAX_TRIES = int(1e5)


def _rng(rng):
  return np.random if rng is None else rng


class Continuous(object):
  """Uniform on [minval, maxval); samples carry `dtype` (float32 by default, like the reference)."""

  def __init__(self, key, minval, maxval, dtype='float32'):
    self.key, self.minval, self.maxval, self.dtype = key, minval, maxval, dtype
    self.keys = {key}

  def sample(self, rng=None):
    value = _rng(rng).uniform(self.minval, self.maxval)
    return {self.key: np.asarray(value).astype(self.dtype)}

  def contains(self, spec):
    return bool(self.minval <= spec[self.key] < self.maxval)


class Discrete(object):
  """Categorical over `candidates` (optionally weighted by `probs`)."""

  def __init__(self, key, candidates, probs=None):
    self.key, self.candidates, self.probs = key, list(candidates), probs
    self.keys = {key}

  def sample(self, rng=None):
    idx = _rng(rng).choice(len(self.candidates), p=self.probs)
    return {self.key: self.candidates[idx]}

  def contains(self, spec):
    return spec[self.key] in self.candidates


class Product(object):
  """Independent product of distributions over disjoint factor sets."""

  def __init__(self, components):
    self.components = list(components)
    self.keys = set()
    for c in self.components:
      if self.keys & c.keys:
        raise ValueError('components of a Product must have disjoint keys')
      self.keys |= c.keys

  def sample(self, rng=None):
    out = {}
    for c in self.components:
      out.update(c.sample(rng=rng))
    return out

  def contains(self, spec):
    return all(c.contains(spec) for c in self.components)


class Mixture(object):
  """Mixture of distributions over the same factor set."""

  def __init__(self, components, probs=None):
    self.components, self.probs = list(components), probs
    self.keys = set(self.components[0].keys)
    if any(c.keys != self.keys for c in self.components):
      raise ValueError('components of a Mixture must have the same keys')

  def sample(self, rng=None):
    idx = _rng(rng).choice(len(self.components), p=self.probs)
    return self.components[idx].sample(rng=rng)

  def contains(self, spec):
    return any(c.contains(spec) for c in self.components)


class Intersection(object):
  """Samples of `components[index_for_sampling]` contained in every other component."""

  def __init__(self, components, index_for_sampling=0):
    self.components, self.index = list(components), index_for_sampling
    self.keys = set(self.components[0].keys)

  def sample(self, rng=None):
    for _ in range(_MAX_TRIES):
      s = self.components[self.index].sample(rng=rng)
      if self.contains(s):
        return s
    raise ValueError('max_tries exceeded when sampling from an Intersection')

  def contains(self, spec):
    return all(c.contains(spec) for c in self.components)


class SetMinus(object):
  """`base` with the support of `hold_out` removed (rejection sampling)."""

  def __init__(self, base, hold_out):
    if not hold_out.keys <= base.keys:
      raise ValueError('keys of hold_out must be a subset of those of base')
    self.base, self.hold_out = base, hold_out
    self.hold_in = base
    self.keys = set(base.keys)

  def sample(self, rng=None):
    for _ in range(_MAX_TRIES):
      s = self.hold_in.sample(rng=rng)
      if not self.hold_out.contains(s):
        return s
    raise ValueError('max_tries exceeded when sampling from a SetMinus')

  def contains(self, spec):
    return self.hold_in.contains(spec) and not self.hold_out.contains(spec)

"""Generators of sprite lists for `init_sprites` (host side, reset time only).

Mirrors of the reference helpers (reference: spriteworld/sprite_generators.py:27-45
generate_sprites, :48-70 chain_generators, :73-98 sample_generator, :101-128
shuffle).  Each returns a zero-argument callable producing a fresh list of
`sprite.Sprite`, back-to-front.
"""
import numpy as np

from spriteworld_amd import sprite


def generate_sprites(factor_dist, num_sprites=1):
  """`num_sprites` (int or callable) sprites drawn from `factor_dist`."""

  def _generate():
    n = num_sprites() if callable(num_sprites) else num_sprites
    return [sprite.Sprite(**factor_dist.sample()) for _ in range(n)]

  return _generate


def chain_generators(*generators):
  """Concatenates the outputs of several generators."""

  def _generate():
    out = []
    for g in generators:
      out.extend(g())
    return out

  return _generate


def sample_generator(generators, p=None):
  """Each call uses one of `generators`, chosen at random."""

  def _generate():
    return generators[np.random.choice(len(generators), p=p)]()

  return _generate


def shuffle(generator):
  """Randomises the z-order of the generated sprites."""

  def _generate():
    sprites = generator()
    order = np.arange(len(sprites))
    np.random.shuffle(order)
    return [sprites[i] for i in order]

  return _generate

"""Renderer descriptors.

Mirrors of the reference constructors (reference:
spriteworld/renderers/pil_renderer.py:30-65 PILRenderer,
spriteworld/renderers/handcrafted.py:115-131 Success,
spriteworld/renderers/color_maps.py:26-28 hsv_to_rgb).  Rasterisation itself is
the HIP render kernel; a PILRenderer here only fixes image size, anti-aliasing
factor, background and the colour map used when an episode pool is lowered.
"""
import colorsys

import numpy as np

from spriteworld_amd import dm_env_compat as dm_env

specs = dm_env.specs


def hsv_to_rgb(c):
  """(h, s, v) -> uint8 (r, g, b), truncating like the reference colour map."""
  return tuple((255 * np.array(colorsys.hsv_to_rgb(*c))).astype(np.uint8))


class color_maps(object):  # namespace, so `renderers.color_maps.hsv_to_rgb` resolves
  hsv_to_rgb = staticmethod(hsv_to_rgb)


class PILRenderer(object):
  """uint8 RGB frame of `image_size`, drawn at `anti_aliasing`x and LANCZOS-shrunk."""

  def __init__(self, image_size=(64, 64), anti_aliasing=1, bg_color=None, color_to_rgb=None):
    self._image_size = tuple(image_size)
    self._anti_aliasing = anti_aliasing
    self._canvas_size = (anti_aliasing * image_size[0], anti_aliasing * image_size[1])
    if color_to_rgb is None:
      color_to_rgb = lambda x: x
    self._color_to_rgb = color_to_rgb
    if bg_color is None:
      bg_color = (0, 0, 0)
    self._bg_color = tuple(bg_color)
    self._observation_spec = specs.Array(shape=self._image_size + (3,), dtype=np.uint8)

  def observation_spec(self):
    return self._observation_spec


class Success(object):
  """Observation key holding task.success() (bool per environment)."""

  def observation_spec(self):
    return specs.Array(shape=(), dtype=np.bool_)


class SpriteFactors(object):
  """Observation of the sprites' factors (mirror of the reference renderer's constructor).

  Batched form: a float64 device tensor [N, S, len(factors)] in the order of `factors`, `shape`
  as its ShapeType value; rows of sprite slots beyond an environment's sprite count are zero
  (the reference returns a list of per-sprite dicts of Python floats)."""

  def __init__(self, factors=None):
    from spriteworld_amd import sprite as sprite_lib
    factors = sprite_lib.FACTOR_NAMES if factors is None else tuple(factors)
    if not set(factors).issubset(set(sprite_lib.FACTOR_NAMES)):
      raise ValueError('Factors have to belong to {}.'.format(sprite_lib.FACTOR_NAMES))
    self._factors = factors
    self._per_object_spec = {f: specs.Array(shape=(), dtype=np.float32) for f in factors}

  def observation_spec(self):
    return self._per_object_spec

"""Host-side sprite record.

Mirror of the reference's `Sprite` constructor and read-only properties
(reference: spriteworld/sprite.py:56-94 constructor, :140-214 properties).  On the
GPU a sprite is one column of the structure-of-arrays state; this class only
carries the ten factors from `init_sprites()` to `lowering.lower_episodes`, which
also accepts the reference's own Sprite objects (duck-typed on the same
attribute names).  Geometry (centred path, vertices, contains_point) lives in
the HIP kernels, not here.
"""
import collections

import numpy as np

from spriteworld_amd import shapes

FACTOR_NAMES = ('x', 'y', 'shape', 'angle', 'scale', 'c0', 'c1', 'c2', 'x_vel', 'y_vel')


class Sprite(object):
  """Ten-factor sprite description (x, y in [0, 1], origin bottom-left)."""

  def __init__(self, x=0.5, y=0.5, shape='square', angle=0, scale=0.1, c0=0, c1=0, c2=0,
               x_vel=0.0, y_vel=0.0):
    if shape not in shapes.SHAPES:
      raise KeyError(shape)
    # Like the reference, the position array takes the dtype of what it is
    # given: np.float32 samples give a float32 array, Python floats float64.
    self._position = np.array([x, y])
    self._shape = shape
    self._angle = angle
    self._scale = scale
    self._color = (c0, c1, c2)
    self._velocity = (x_vel, y_vel)

  x = property(lambda self: self._position[0])
  y = property(lambda self: self._position[1])
  shape = property(lambda self: self._shape)
  angle = property(lambda self: self._angle)
  scale = property(lambda self: self._scale)
  c0 = property(lambda self: self._color[0])
  c1 = property(lambda self: self._color[1])
  c2 = property(lambda self: self._color[2])
  x_vel = property(lambda self: self._velocity[0])
  y_vel = property(lambda self: self._velocity[1])
  color = property(lambda self: self._color)
  position = property(lambda self: self._position)
  velocity = property(lambda self: self._velocity)

  @property
  def factors(self):
    return collections.OrderedDict((name, getattr(self, name)) for name in FACTOR_NAMES)

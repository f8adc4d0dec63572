"""Host-side sprite record.

Mirror of the reference's `Sprite` constructor and read-only properties
(reference: spriteworld/sprite.py:56-94 constructor, :140-214 properties).  On the
GPU a sprite is one column of the structure-of-arrays state; this class only
carries the ten factors from `init_sprites()` to `lowering.lower_episodes`, which
also accepts the reference's own Sprite objects (duck-typed on the same
attribute names).  Geometry (centred path, vertices, contains_point) lives in
the HIP kernels, not here.

`LiveSprite` is the handle on a sprite of a RUNNING environment: its attribute
setters are the reference's (:152-175 -- shape resets the centred path, angle
and scale transform it incrementally, scale by the DIFFERENCE (s - old)) and act
on the device state through `swb_set_sprite_attr`.
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


_MAX_TRIES = int(1e6)       # sprite.py:35


def point_in_centered_path(path, tx, ty):
  """matplotlib `_path.h` point_in_path_impl (radius 0, closed polygon) for one point: crossing parity with the
  comparisons and products in matplotlib's order, all float64 (numpy scalars: IEEE, no FMA)."""
  path = np.asarray(path, dtype=np.float64)
  n = len(path)
  tx, ty = np.float64(tx), np.float64(ty)
  inside = False
  vx0, vy0 = path[0]
  yflag0 = vy0 >= ty
  for i in range(1, n + 1):
    vx1, vy1 = path[i % n]
    yflag1 = vy1 >= ty
    if yflag0 != yflag1:
      if (((vy1 - ty) * (vx0 - vx1)) >= ((vx1 - tx) * (vy0 - vy1))) == yflag1:
        inside = not inside
    yflag0, vx0, vy0 = yflag1, vx1, vy1
  return bool(inside)


class LiveSprite(object):
  """Sprite `index` of environment `env` of a running engine, in its current episode.

  Read access mirrors the reference's properties (sprite.py:140-214); `shape`, `angle` and `scale` can be
  assigned exactly like on the reference's Sprite (sprite.py:152-175): the change shows in the next
  observation(), hit-test and SpriteFactors read, and ends with the episode (a reset draws fresh sprites).
  Position, colour and velocity are read-only properties, as in the reference (:140-204)."""

  def __init__(self, environment, env, index):
    self._environment, self._env, self._index = environment, int(env), int(index)

  def _read(self):
    return self._environment.engine.get_sprite(self._env, self._index)

  def _factor_row(self):
    return self._environment.engine.factors()[self._env, self._index].cpu().numpy()

  x = property(lambda self: self._factor_row()[0])
  y = property(lambda self: self._factor_row()[1])
  c0 = property(lambda self: self._factor_row()[5])
  c1 = property(lambda self: self._factor_row()[6])
  c2 = property(lambda self: self._factor_row()[7])
  x_vel = property(lambda self: self._factor_row()[8])
  y_vel = property(lambda self: self._factor_row()[9])
  position = property(lambda self: self._factor_row()[:2])
  color = property(lambda self: tuple(self._factor_row()[5:8]))
  velocity = property(lambda self: tuple(self._factor_row()[8:10]))

  @property
  def shape(self):
    return shapes.SHAPE_NAMES[self._read()['shape']]

  @shape.setter
  def shape(self, s):
    if s not in shapes.SHAPES:
      raise KeyError(s)
    self._environment.set_sprite_attr(self._env, self._index, 'shape', s)

  @property
  def angle(self):
    return self._read()['angle']

  @angle.setter
  def angle(self, a):
    self._environment.set_sprite_attr(self._env, self._index, 'angle', a)

  @property
  def scale(self):
    return self._read()['scale']

  @scale.setter
  def scale(self, s):
    self._environment.set_sprite_attr(self._env, self._index, 'scale', s)

  @property
  def centered_path(self):
    """f64[n, 2]: the sprite's centred path as the engine holds it (sprite.py:96-101; swb_get_sprite)."""
    return self._read()['path']

  @property
  def out_of_frame(self):
    """sprite.py:135-138."""
    pos = self.position
    return not (np.all(pos >= [0., 0.]) and np.all(pos <= [1., 1.]))

  def contains_point(self, point):
    """sprite.py:113-115 -> matplotlib Path.contains_point (radius 0): the even-odd rule of `_path.h` point_in_path on the
    centred path, float64, in matplotlib's operation order (the hit-test the cover kernel's `contains_point_wave` and the
    oracle's `point_in_centered_path` restate)."""
    t = np.asarray(point, dtype=np.float64) - self.position
    return point_in_centered_path(self.centered_path, t[0], t[1])

  def sample_contained_position(self):
    """sprite.py:117-126: rejection sampling in the centred path's bounding box, consuming numpy's global stream exactly
    as the reference does (one `np.random.uniform(low, high)` pair per try)."""
    path, pos = self.centered_path, self.position
    low, high = np.min(path, axis=0), np.max(path, axis=0)
    for _ in range(_MAX_TRIES):
      sample = pos + np.random.uniform(low, high)
      if point_in_centered_path(path, sample[0] - pos[0], sample[1] - pos[1]):
        return sample
    raise ValueError('max_tries exceeded. There is almost surely an error in the SpriteWorld library code.')

  @property
  def vertices(self):
    """sprite.py:128-133: the centred path translated by the position ((1*x + 0*y) + px, as matplotlib does)."""
    path = self._read()['path']
    pos = self._factor_row()[:2]
    return np.stack([1.0 * path[:, 0] + 0.0 * path[:, 1] + pos[0], 0.0 * path[:, 0] + 1.0 * path[:, 1] + pos[1]], axis=1)

  @property
  def factors(self):
    row = self._factor_row()
    d = collections.OrderedDict(zip(FACTOR_NAMES, row.tolist()))
    d['shape'] = shapes.SHAPE_NAMES[int(row[2]) - 1]
    return d

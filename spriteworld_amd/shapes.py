"""Unit-area vertex tables for the twelve Spriteworld shapes.

Host-side mirror of the reference's shape library (reference:
spriteworld/shapes.py:34-116 `polygon`, `star`, `spokes`;
spriteworld/constants.py:27-56 `SHAPES`, `ShapeType`).  The tables are uploaded
to the GPU once (`swb_upload_shapes`) and every vertex the kernels touch is
derived from them in float64, so the values have to be bit-identical to the
reference's: tests/test_golden.py::test_shape_tables_match_reference_values checks
that against the committed golden table (tests/golden/shapes.json) and, when the
reference tree is present, against `spriteworld.constants.SHAPES` itself.
"""
import enum

import numpy as np


def _unit_circle_points(angles):
  """Rows (cos t, sin t) for a 1-D array of angles."""
  return np.stack([np.cos(angles), np.sin(angles)], axis=1)


def polygon(num_sides, theta_0=0.):
  """Regular `num_sides`-gon with a vertex at angle `theta_0`, area 1."""
  theta = 2 * np.pi / num_sides
  angles = np.array([i * theta + theta_0 for i in range(num_sides)])
  area = num_sides * np.sin(theta / 2) * np.cos(theta / 2)
  return (1 * _unit_circle_points(angles)) / np.sqrt(area)


def star(num_sides, point_height=1, theta_0=0.):
  """Regular star: inner vertices at radius 1, tips at 1 + point_height."""
  tip_radius = 1 + point_height
  theta = 2 * np.pi / num_sides
  verts = np.empty([2 * num_sides, 2])
  inner = np.array([i * theta + theta_0 for i in range(num_sides)])
  tips = np.array([(i + 0.5) * theta + theta_0 for i in range(num_sides)])
  verts[0::2] = 1 * _unit_circle_points(inner)
  verts[1::2] = tip_radius * _unit_circle_points(tips)
  area = tip_radius * num_sides * np.sin(theta / 2)
  return verts / np.sqrt(area)


def spokes(num_sides, spoke_height=1, theta_0=0.):
  """Star with rectangular points ("spokes")."""
  theta = 2 * np.pi / num_sides
  verts = np.empty([3 * num_sides, 2])
  base = np.array([i * theta + theta_0 for i in range(num_sides)])
  before = np.array([(i - 0.5) * theta + theta_0 for i in range(num_sides)])
  before[0] = -0.5 * theta + theta_0
  after = np.array([(i + 0.5) * theta + theta_0 for i in range(num_sides)])
  corner = 1 * _unit_circle_points(base)
  verts[0::3] = spoke_height * _unit_circle_points(before) + corner
  verts[1::3] = corner
  verts[2::3] = spoke_height * _unit_circle_points(after) + corner
  area = num_sides * np.sin(theta / 2) * (2 + np.cos(theta / 2))
  return verts / np.sqrt(area)


# Name -> (n, 2) float64 vertex array, in ShapeType order.
SHAPES = {
    'triangle': polygon(num_sides=3, theta_0=np.pi / 2),
    'square': polygon(num_sides=4, theta_0=np.pi / 4),
    'pentagon': polygon(num_sides=5, theta_0=np.pi / 2),
    'hexagon': polygon(num_sides=6),
    'octagon': polygon(num_sides=8),
    'circle': polygon(num_sides=30),
    'star_4': star(num_sides=4, theta_0=np.pi / 4),
    'star_5': star(num_sides=5, theta_0=np.pi + np.pi / 10),
    'star_6': star(num_sides=6),
    'spoke_4': spokes(num_sides=4, theta_0=np.pi / 4),
    'spoke_5': spokes(num_sides=5, theta_0=np.pi + np.pi / 10),
    'spoke_6': spokes(num_sides=6),
}


class ShapeType(enum.IntEnum):
  """Integer ids of SHAPES (1-based, as the reference's SpriteFactors emits)."""
  triangle = 1
  square = 2
  pentagon = 3
  hexagon = 4
  octagon = 5
  circle = 6
  star_4 = 7
  star_5 = 8
  star_6 = 9
  spoke_4 = 10
  spoke_5 = 11
  spoke_6 = 12


SHAPE_NAMES = tuple(t.name for t in ShapeType)


def shape_index(name):
  """0-based row of `name` in the packed device table."""
  return ShapeType[name].value - 1


def packed_table(shapes=None):
  """(vertices f64 [total, 2], offsets i32 [n_shapes + 1]) in ShapeType order."""
  shapes = SHAPES if shapes is None else shapes
  verts = [np.asarray(shapes[name], dtype=np.float64) for name in SHAPE_NAMES]
  offsets = np.zeros(len(verts) + 1, dtype=np.int32)
  offsets[1:] = np.cumsum([len(v) for v in verts])
  return np.ascontiguousarray(np.concatenate(verts, axis=0)), offsets

"""Lowering of reference-style Python objects to the C ABI (include/swb.h).

The reference's configuration surface is duck-typed Python: `get_config()`
returns {'task', 'action_space', 'renderers', 'init_sprites', ...} (reference:
spriteworld/configs/cobra/goal_finding_more_distractors.py:41-96) and
`Environment.__init__` just stores them (spriteworld/environment.py:34-72).
This module reads the very attributes those objects set -- whether they are the
reference's own classes or the mirrors in this package -- and produces

  * `SwbConfig`   the flat struct consumed by `swb_create` (and by the oracle);
  * `Pool`        a pool of initial states ("what init_sprites() returned"),
                  with everything that is constant during an episode resolved
                  on the host: cos/sin of the angle, RGB fill colour, and the
                  task membership label of each sprite (tasks.py:134-137,
                  196-205 evaluate `contains(sprite.factors)` every step, but
                  `step()` only ever changes x and y, so for factor filters that
                  do not key on x/y the result is fixed at reset; for filters that
                  do, the label is tabulated over the cells of the grid their
                  interval bounds cut the plane into and looked up every step).
"""
import ctypes as C
import math

import numpy as np

from spriteworld_amd import _abi
from spriteworld_amd import shapes as _shapes


class LoweringError(ValueError):
  pass


def _cls(obj):
  return type(obj).__name__


# --------------------------------------------------------------------------- #
# Config                                                                       #
# --------------------------------------------------------------------------- #
def _lower_task(task, out, pos_is_f32=True):
  """Fills one SwbTask from a FindGoalPosition / Clustering / NoReward object."""
  name = _cls(task)
  out.n_xcuts = out.n_ycuts = 0
  if name == 'NoReward':
    out.kind = _abi.TASK_NO_REWARD
  elif name == 'FindGoalPosition':
    out.kind = _abi.TASK_FIND_GOAL
    goal = np.asarray(task._goal_position, dtype=np.float64)
    weights = np.asarray(task._weights_dimensions, dtype=np.float64)
    if goal.shape != (2,) or weights.shape != (2,):
      raise LoweringError('goal_position / weights_dimensions must have 2 entries')
    out.goal_position[0], out.goal_position[1] = goal
    out.weights_dimensions[0], out.weights_dimensions[1] = weights
    out.terminate_distance = float(task._terminate_distance)
    out.raw_reward_multiplier = float(task._raw_reward_multiplier)
    out.terminate_bonus = float(task._terminate_bonus)
    out.sparse_reward = int(bool(task._sparse_reward))
  elif name == 'Clustering':
    out.kind = _abi.TASK_CLUSTERING
    out.termination_threshold = float(task._termination_threshold)
    out.terminate_bonus = float(task._terminate_bonus)
    out.reward_range = float(task._reward_range)
    out.sparse_reward = int(bool(task._sparse_reward))
  else:
    raise LoweringError('unsupported task type: ' + name)
  if name in ('FindGoalPosition', 'Clustering'):
    xcuts, ycuts = position_cuts(task, pos_is_f32)
    out.n_xcuts, out.n_ycuts = len(xcuts), len(ycuts)
    for k, v in enumerate(xcuts):
      out.xcuts[k] = v
    for k, v in enumerate(ycuts):
      out.ycuts[k] = v


_AGG = {'nansum': _abi.AGG_SUM, 'nanmax': _abi.AGG_MAX, 'nanmin': _abi.AGG_MIN,
        'nanmean': _abi.AGG_MEAN}
_TERM = {'all': _abi.TERM_ALL, 'any': _abi.TERM_ANY}


_POSITION_KEYS = ('x', 'y')


def _task_distribs(sub):
  name = _cls(sub)
  if name == 'FindGoalPosition':
    return [sub._filter_distrib] if sub._filter_distrib is not None else []
  if name == 'Clustering':
    return list(sub._cluster_distribs)
  return []


def _collect_position_bounds(d, out):
  """Every bound of an interval test on x / y in a factor-distribution tree (the reference's classes or the mirrors):
  Continuous.contains is `minval <= v < maxval` (factor_distributions.py:105-112); Product / Mixture / Intersection hold
  `components`, SetMinus `base` + `hold_out`, Selection `base` + `filtering`."""
  if d is None:
    return
  key = getattr(d, 'key', None)
  if key is not None and hasattr(d, 'minval') and hasattr(d, 'maxval'):
    if key in _POSITION_KEYS:
      out[key].extend([d.minval, d.maxval])
    return
  if key is not None and hasattr(d, 'candidates'):
    if key in _POSITION_KEYS:
      raise LoweringError('a Discrete distribution over %r in a task filter (membership by equality with a position) is not '
                          'supported' % key)
    return
  kids = list(getattr(d, 'components', ()) or ())
  for attr in ('base', 'hold_out', 'filtering'):
    if getattr(d, attr, None) is not None:
      kids.append(getattr(d, attr))
  if not kids:
    keys = getattr(d, 'keys', None)
    keys = set(keys() if callable(keys) else (keys or ()))
    if keys & set(_POSITION_KEYS):
      raise LoweringError('task filter of type %s keys on position and is not a composition of Continuous distributions' % _cls(d))
  for k in kids:
    _collect_position_bounds(k, out)


def _threshold(bound, pos_is_f32):
  """The smallest position value v (a float32 when positions are float32) for which numpy's `v >= bound` holds -- `v < bound`
  is its negation under the same rule.  NEP 50: a Python number compared with an np.float32 is first cast to float32; an
  np.float64 bound promotes the comparison to float64."""
  if not pos_is_f32:
    return float(bound)
  if isinstance(bound, (np.floating, np.ndarray)) and np.asarray(bound).dtype == np.float64:
    b = np.float64(bound)
    t = np.float32(b)
    if np.float64(t) < b:
      t = np.nextafter(t, np.float32(np.inf), dtype=np.float32)
    return float(t)
  return float(np.float32(bound))


def position_cuts(sub, pos_is_f32):
  """(xcuts, ycuts): the ascending thresholds that the position tests of a (sub-)task's filter / cluster distributions cut
  the two axes at; ([], []) when it does not key on position.  See swb_task in include/swb.h."""
  out = {'x': [], 'y': []}
  for d in _task_distribs(sub):
    _collect_position_bounds(d, out)
  cuts = []
  for key in _POSITION_KEYS:
    vals = sorted(set(_threshold(b, pos_is_f32) for b in out[key]))
    vals = [v for v in vals if not math.isnan(v)]
    if len(vals) > _abi.SWB_MAX_CUTS:
      raise LoweringError('task filter tests %s against %d different bounds (at most %d supported)' % (key, len(vals), _abi.SWB_MAX_CUTS))
    cuts.append(vals)
  return cuts[0], cuts[1]


def _label_from_factors(sub, factors):
  name = _cls(sub)
  if name == 'FindGoalPosition':
    f = sub._filter_distrib
    return int(f is None or bool(f.contains(factors)))
  if name == 'Clustering':
    for ci, distrib in enumerate(sub._cluster_distribs):
      if distrib.contains(factors):
        return ci
    return -1
  return 0


def cell_labels_of(sub, sprite, xcuts, ycuts, pos_dtype):
  """i8[SWB_MAX_CELLS]: the sprite's label in every cell of the task's position grid -- the reference's own
  `contains(sprite.factors)` evaluated with x / y at a representative of the cell (its lower-left corner; -inf below the
  first cut), the other factors as the sprite holds them.  Self-checked against direct evaluation at random positions."""
  cells = np.zeros(_abi.SWB_MAX_CELLS, np.int8)
  factors = dict(sprite.factors)
  reps_x = [pos_dtype(-np.inf)] + [pos_dtype(v) for v in xcuts]
  reps_y = [pos_dtype(-np.inf)] + [pos_dtype(v) for v in ycuts]
  nx = len(xcuts) + 1
  for cy, vy in enumerate(reps_y):
    for cx, vx in enumerate(reps_x):
      factors['x'], factors['y'] = vx, vy
      cells[cy * nx + cx] = _label_from_factors(sub, factors)
  rng = np.random.RandomState(12345)
  for _ in range(16):
    vx, vy = pos_dtype(rng.uniform(-0.2, 1.2)), pos_dtype(rng.uniform(-0.2, 1.2))
    factors['x'], factors['y'] = vx, vy
    cx = sum(1 for c in xcuts if float(vx) >= c)
    cy = sum(1 for c in ycuts if float(vy) >= c)
    if _label_from_factors(sub, factors) != cells[cy * nx + cx]:
      raise LoweringError('task filter is not a function of the position grid its Continuous bounds define (at x=%r, y=%r)' % (vx, vy))
  return cells


def subtasks_of(task):
  """[sub-tasks] of a MetaAggregated task, or [task]."""
  return list(task._subtasks) if _cls(task) == 'MetaAggregated' else [task]


def find_pil_renderer(renderers):
  """(name, renderer) of the PILRenderer in a `renderers` dict, or (None, None)."""
  for name, r in renderers.items():
    if hasattr(r, '_anti_aliasing') and hasattr(r, '_image_size'):
      return name, r
  return None, None


def _bg_of(renderer):
  if hasattr(renderer, '_bg_color'):
    return tuple(int(v) for v in renderer._bg_color)
  return tuple(int(v) for v in renderer._canvas_bg.getpixel((0, 0)))  # reference PILRenderer


def lower_config(task, action_space, renderers, keep_in_frame=True, max_episode_length=1000,
                 num_envs=1, max_sprites=1, pos_is_f32=True, action_dtype=np.float64):
  """Builds the SwbConfig for Environment(task, action_space, renderers, ...).

  `action_dtype`: np.float64 (what the reference's `action_space.sample()` returns) or
  np.float32 (what its `action_spec()` declares); numpy's promotion rules make the two
  differ in the last bits of motions, click offsets and some rewards, so the engine follows
  whichever the caller uses.  Ignored for Embodied (integer actions)."""
  cfg = _abi.SwbConfig()
  cfg.n_envs = int(num_envs)
  cfg.max_sprites = int(max_sprites)
  if not 1 <= cfg.max_sprites <= _abi.SWB_MAX_SPRITES:
    raise LoweringError('max_sprites must be in [1, %d]' % _abi.SWB_MAX_SPRITES)
  _, pil = find_pil_renderer(renderers)
  if pil is None:  # e.g. tests/configs_test.py builds environments with renderers={}
    cfg.image_h, cfg.image_w, cfg.anti_aliasing = 8, 8, 1
  else:
    cfg.image_h, cfg.image_w = int(pil._image_size[0]), int(pil._image_size[1])
    cfg.anti_aliasing = int(pil._anti_aliasing)
    for i, v in enumerate(_bg_of(pil)[:3]):
      cfg.bg_rgb[i] = v
  # action space
  name = _cls(action_space)
  if name in ('SelectMove', 'DragAndDrop'):
    cfg.action_space = (_abi.ACTION_SELECT_MOVE if name == 'SelectMove' else
                        _abi.ACTION_DRAG_AND_DROP)
    cfg.action_scale = float(action_space._scale)
    cfg.motion_cost = float(action_space._motion_cost)
    # _noise_scale is not lowered: the noise is added to the action tensor by the caller
    # (BatchedEnvironment.step does it on the device), as the reference adds it before anything else
  elif name == 'Embodied':
    cfg.action_space = _abi.ACTION_EMBODIED
    cfg.action_scale = float(action_space._step_size)
    cfg.motion_cost = float(action_space._motion_cost)
  else:
    raise LoweringError('unsupported action space: ' + name)
  if np.dtype(action_dtype) not in (np.dtype(np.float32), np.dtype(np.float64)):
    raise LoweringError('action_dtype must be float32 or float64')
  cfg.action_is_f32 = int(np.dtype(action_dtype) == np.dtype(np.float32) and
                          cfg.action_space != _abi.ACTION_EMBODIED)
  cfg.keep_in_frame = int(bool(keep_in_frame))
  cfg.max_episode_length = int(max_episode_length)
  cfg.pos_is_f32 = int(bool(pos_is_f32))
  # task(s)
  subs = subtasks_of(task)
  if not 1 <= len(subs) <= _abi.SWB_MAX_TASKS:
    raise LoweringError('between 1 and %d sub-tasks supported' % _abi.SWB_MAX_TASKS)
  cfg.n_tasks = len(subs)
  if _cls(task) == 'MetaAggregated':
    cfg.is_meta = 1
    agg = getattr(task._reward_aggregator, '__name__', '')
    term = getattr(task._termination_criterion, '__name__', '')
    if agg not in _AGG or term not in _TERM:
      raise LoweringError('unknown MetaAggregated aggregator/criterion: %s/%s' % (agg, term))
    cfg.meta_aggregator, cfg.meta_termination = _AGG[agg], _TERM[term]
    cfg.meta_terminate_bonus = float(task._terminate_bonus)
  for i, sub in enumerate(subs):
    _lower_task(sub, cfg.tasks[i], pos_is_f32=bool(pos_is_f32))
  return cfg


# --------------------------------------------------------------------------- #
# Pool                                                                         #
# --------------------------------------------------------------------------- #
class Pool(object):
  """Host arrays of a reset pool (layout of `swb_pool`, include/swb.h)."""

  FIELDS = ('n_sprites', 'x', 'y', 'x_vel', 'y_vel', 'scale', 'cos_a', 'sin_a', 'shape', 'rgb',
            'label', 'pool_base', 'pool_len')

  def __init__(self, n_entries, max_sprites, n_tasks):
    P, S, T = int(n_entries), int(max_sprites), int(n_tasks)
    self.n_entries, self.max_sprites, self.n_tasks = P, S, T
    self.n_sprites = np.zeros(P, np.int32)
    for name in ('x', 'y', 'x_vel', 'y_vel', 'scale', 'cos_a', 'sin_a'):
      setattr(self, name, np.zeros((P, S), np.float64))
    self.scale[:] = 1.0
    self.cos_a[:] = 1.0
    self.shape = np.zeros((P, S), np.int32)
    self.rgb = np.zeros((P, S, 4), np.uint8)
    self.label = np.zeros((P, T, S), np.int8)
    self.cell_label = None      # i8[P, T, S, SWB_MAX_CELLS] when a task keys on position (swb_task::n_xcuts), else None
    self.pool_base = None
    self.pool_len = None
    # Not consumed by the kernels; kept for SpriteFactors-style observations.
    self.angle = np.zeros((P, S), np.float64)
    self.color = np.zeros((P, S, 3), np.float64)
    # Host-only: bit 0 / 1 set when the sprite's angle / scale is an np.float32 (factor distributions draw float32).
    # The reference's setters take `a - self._angle` / `s - self._scale` in that type (sprite.py:163,173 under NEP 50).
    self.attr_f32 = np.zeros((P, S), np.uint8)

  def assign_round_robin(self, num_envs, entries_per_env=None):
    """Env n draws entries [n*k, (n+1)*k) cyclically (k = P // num_envs)."""
    k = entries_per_env or max(self.n_entries // num_envs, 1)
    if k * num_envs > self.n_entries and entries_per_env:
      raise LoweringError('pool too small for %d envs x %d entries' % (num_envs, k))
    self.pool_base = ((np.arange(num_envs, dtype=np.int64) * k) % max(self.n_entries - k + 1, 1)
                      ).astype(np.int32)
    self.pool_len = np.full(num_envs, k, np.int32)
    return self

  def as_struct(self):
    if self.pool_base is None:
      raise LoweringError('pool has no env assignment (call assign_round_robin)')
    for name in self.FIELDS:
      a = getattr(self, name)
      if not a.flags['C_CONTIGUOUS']:
        setattr(self, name, np.ascontiguousarray(a))
    s = _abi.SwbPool()
    s.n_entries = self.n_entries
    for name in self.FIELDS:
      setattr(s, name, getattr(self, name).ctypes.data)
    self.angle = np.ascontiguousarray(self.angle, dtype=np.float64)
    self.color = np.ascontiguousarray(self.color, dtype=np.float64)
    s.angle, s.color = self.angle.ctypes.data, self.color.ctypes.data
    self.attr_f32 = np.ascontiguousarray(self.attr_f32, dtype=np.uint8)
    s.attr_f32 = self.attr_f32.ctypes.data
    if self.cell_label is not None:
      self.cell_label = np.ascontiguousarray(self.cell_label, dtype=np.int8)
      assert self.cell_label.shape == (self.n_entries, self.n_tasks, self.max_sprites, _abi.SWB_MAX_CELLS)
      s.cell_label = self.cell_label.ctypes.data
    return s


def _label_of(sub, sprite):
  name = _cls(sub)
  if name == 'FindGoalPosition':
    f = sub._filter_distrib
    return int(f is None or bool(f.contains(sprite.factors)))
  if name == 'Clustering':
    for ci, distrib in enumerate(sub._cluster_distribs):
      if distrib.contains(sprite.factors):
        return ci
    return -1
  return 0


def position_dtype(episodes):
  """np.float32 / np.float64: dtype of the sprites' position arrays (must agree)."""
  kinds = set()
  for ep in episodes:
    for s in ep:
      kinds.add(np.asarray(s.position).dtype)
  if not kinds:
    return np.dtype(np.float32)
  if len(kinds) > 1:
    raise LoweringError('sprites mix position dtypes %s' % sorted(map(str, kinds)))
  dt = kinds.pop()
  if dt not in (np.dtype(np.float32), np.dtype(np.float64)):
    raise LoweringError('unsupported position dtype %s' % dt)
  return dt


def lower_episodes(episodes, task, renderers, max_sprites=None):
  """List of sprite lists (one per reset, back-to-front order) -> Pool."""
  subs = subtasks_of(task)
  pos_dtype = position_dtype(episodes).type
  cuts = [position_cuts(sub, pos_dtype == np.float32) for sub in subs]
  keyed = [bool(xc or yc) for xc, yc in cuts]
  _, pil = find_pil_renderer(renderers)
  to_rgb = pil._color_to_rgb if pil is not None else (lambda c: (0, 0, 0))
  S = max_sprites or max([len(ep) for ep in episodes] + [1])
  pool = Pool(len(episodes), S, len(subs))
  if any(keyed):
    pool.cell_label = np.zeros((len(episodes), len(subs), S, _abi.SWB_MAX_CELLS), np.int8)
  for e, ep in enumerate(episodes):
    if len(ep) > S:
      raise LoweringError('episode %d has %d sprites > max_sprites %d' % (e, len(ep), S))
    pool.n_sprites[e] = len(ep)
    for s, sp in enumerate(ep):
      pos = np.asarray(sp.position)
      pool.x[e, s], pool.y[e, s] = float(pos[0]), float(pos[1])
      pool.x_vel[e, s], pool.y_vel[e, s] = float(sp.velocity[0]), float(sp.velocity[1])
      pool.scale[e, s] = float(sp.scale)
      theta = math.radians(sp.angle)  # matplotlib Affine2D.rotate_deg
      pool.cos_a[e, s], pool.sin_a[e, s] = math.cos(theta), math.sin(theta)
      pool.angle[e, s] = float(sp.angle)
      # (np.float32 scalars and the 0-d float32 arrays the reference's Continuous.sample() returns alike)
      pool.attr_f32[e, s] = ((1 if getattr(sp.angle, 'dtype', None) == np.float32 else 0) |
                             (2 if getattr(sp.scale, 'dtype', None) == np.float32 else 0))
      pool.shape[e, s] = _shapes.shape_index(sp.shape)
      rgb = to_rgb(sp.color)
      pool.rgb[e, s, :3] = np.asarray(rgb).astype(np.uint8)
      pool.color[e, s] = [float(c) for c in sp.color]
      for t, sub in enumerate(subs):
        pool.label[e, t, s] = _label_of(sub, sp)
        if keyed[t]:       # membership by position: the label in every cell of the task's grid (looked up every step)
          pool.cell_label[e, t, s] = cell_labels_of(sub, sp, cuts[t][0], cuts[t][1], pos_dtype)
  return pool

"""The sprite-setter test scenarios (reference sprite.py:152-175), written against an
engine FACTORY so that tests/test_gpu_setters.py runs them on the HIP engine (through the C ABI, the OV builds of
the step kernel) and tests/test_sprite_setters.py runs the same scripts on CPU against tests/_fake_engine.py (which
checks the scripts and the host-side API, not the kernels)."""
import numpy as np
import pytest

from spriteworld_amd import _abi, shapes, workloads


def _bits(a):
  return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def _compare(t, ora, eng, want, got):
  st_o, st_g = ora.state(), eng.state()
  assert not got['error'].any(), (t, np.flatnonzero(got['error'])[:8])
  np.testing.assert_array_equal(got['step_type'], want['step_type'], err_msg='step_type t=%d' % t)
  np.testing.assert_array_equal(_bits(st_g['x']), _bits(st_o['x']), err_msg='x t=%d' % t)
  np.testing.assert_array_equal(_bits(st_g['y']), _bits(st_o['y']), err_msg='y t=%d' % t)
  for k in ('step_count', 'reset_next', 'episode', 'pool_entry', 'n_sprites'):
    np.testing.assert_array_equal(st_g[k], st_o[k], err_msg='%s t=%d' % (k, t))
  np.testing.assert_array_equal(got['success'], want['success'], err_msg='success t=%d' % t)
  gr, wr = got['reward'], want['reward']
  assert np.array_equal(np.isnan(gr), np.isnan(wr)), 'reward NaN pattern t=%d' % t
  ok = ~np.isnan(wr)
  np.testing.assert_array_equal(_bits(gr[ok]), _bits(wr[ok]), err_msg='reward t=%d' % t)
  diff = np.abs(got['obs'].astype(np.int16) - want['obs'].astype(np.int16))
  assert diff.max() == 0, ('frame diff', int(diff.max()), int((diff > 0).sum()), t, np.argwhere(diff > 0)[:5].tolist())
  return st_o


def run_parity(make_engine, name, n_envs, steps, aa, seed=0, calls_per_step=6):
  from oracle import oracle
  cfg, pool, sample = workloads.build(name, n_envs, episodes_per_env=3, seed=seed, anti_aliasing=aa)
  ora = oracle.Engine(cfg, pool)
  eng = make_engine(cfg, pool)
  rng = np.random.default_rng(seed + 100)
  srng = np.random.RandomState(seed + 5)
  n_shapes = len(shapes.SHAPES)
  applied = 0
  for t in range(steps):
    a = sample(rng)
    want = ora.step(a)
    eng.step(a)
    st = _compare(t, ora, eng, want, eng.outputs_host())
    # setters on live sprites of random environments (never one whose episode just ended)
    live = np.flatnonzero((st['reset_next'] == 0) & (st['n_sprites'] > 0))
    for _ in range(calls_per_step if len(live) else 0):
      env = int(srng.choice(live))
      k = int(srng.randint(0, st['n_sprites'][env]))
      attr = int(srng.randint(0, 3))
      value = (float(srng.randint(0, n_shapes)) if attr == _abi.ATTR_SHAPE else
               float(srng.choice([0., 17., 45., 90., 133.5, 270., 359.])) if attr == _abi.ATTR_ANGLE else
               float(srng.choice([0.08, 0.12, 0.2, 0.3])))
      ora.set_sprite_attr(env, k, attr, value)
      eng.set_sprite_attr(env, k, attr, value)
      applied += 1
      so, sg = ora.get_sprite(env, k), eng.get_sprite(env, k)
      assert (so['shape'], so['angle'], so['scale']) == (sg['shape'], sg['angle'], sg['scale'])
      assert np.array_equal(_bits(so['path']), _bits(sg['path'])), (t, env, k, attr, value)
    # observation() between steps shows the modified sprites at once
    if t % 3 == 0:
      np.testing.assert_array_equal(eng.render().cpu().numpy(), ora.render(), err_msg='render t=%d' % t)
  assert applied > 0
  assert eng.variant()['kernel'].startswith(('swb_resample_kernel', 'swb_fill_kernel', 'none'))
  eng.close()


def factors_and_reset_case(make_engine, error_type):
  """SpriteFactors shows the new attribute values; the overrides end at the environment's next reset; an
  environment without a live episode refuses the call."""
  cfg, pool, sample = workloads.build('goal_s5', 8, episodes_per_env=3, seed=3, anti_aliasing=5)
  eng = make_engine(cfg, pool)
  with pytest.raises(error_type):
    eng.set_sprite_attr(0, 0, _abi.ATTR_ANGLE, 10.0)          # never reset: no sprites yet
  rng = np.random.default_rng(1)
  eng.step(sample(rng))                                        # FIRST
  before = eng.factors().cpu().numpy().copy()
  eng.set_sprite_attr(2, 1, _abi.ATTR_ANGLE, 77.0)
  eng.set_sprite_attr(2, 1, _abi.ATTR_SCALE, 0.31)
  eng.set_sprite_attr(2, 0, _abi.ATTR_SHAPE, float(shapes.shape_index('star_5')))
  after = eng.factors().cpu().numpy()
  want = before.copy()
  want[2, 1, 3], want[2, 1, 4] = 77.0, 0.31
  want[2, 0, 2] = shapes.shape_index('star_5') + 1             # constants.ShapeType value
  np.testing.assert_array_equal(after, want)
  with pytest.raises(error_type):
    eng.set_sprite_attr(2, 9, _abi.ATTR_ANGLE, 10.0)          # no such sprite
  eng.reset_all()
  eng.step(sample(rng))                                        # FIRST again: fresh sprites
  st = eng.state()
  fresh = eng.factors().cpu().numpy()
  e = st['pool_entry'][2]
  assert fresh[2, 1, 3] == pool.angle[e, 1] and fresh[2, 1, 4] == pool.scale[e, 1]
  assert fresh[2, 0, 2] == pool.shape[e, 0] + 1
  eng.close()


def live_sprite_case():
  """`env.sprites(e)[k].angle = a` etc. on a BatchedEnvironment: the handle's vertices equal what matplotlib
  computes for the reference's incremental transforms (sprite.py:96-101,152-175), the SpriteFactors
  observation shows the new values, and a filter keyed on the changed factor re-labels the sprite."""
  from matplotlib import path as mpl_path
  from matplotlib import transforms as mpl_transforms
  from spriteworld_amd import action_spaces, environment, renderers, sprite_generators, tasks
  from spriteworld_amd import factor_distributions as distribs
  np.random.seed(4)
  factors = distribs.Product([
      distribs.Continuous('x', 0.2, 0.8), distribs.Continuous('y', 0.2, 0.8),
      distribs.Discrete('shape', ['square', 'triangle', 'star_5']), distribs.Discrete('scale', [0.15]),
      distribs.Discrete('angle', [30.0]), distribs.Continuous('c0', 0., 1.), distribs.Continuous('c1', 0.5, 1.),
      distribs.Continuous('c2', 0.9, 1.)])
  # targets = circles: none at first, so the (vacuous) success ends every episode after one step; the goal is
  # out of reach for the sprite that becomes a circle below
  env = environment.BatchedEnvironment(
      task=tasks.FindGoalPosition(filter_distrib=distribs.Discrete('shape', ['circle']), goal_position=(3., 3.),
                                  terminate_distance=0.01),
      action_space=action_spaces.SelectMove(scale=0.0),
      renderers={'image': renderers.PILRenderer(image_size=(64, 64), anti_aliasing=5, color_to_rgb=renderers.hsv_to_rgb),
                 'factors': renderers.SpriteFactors()},
      init_sprites=sprite_generators.generate_sprites(factors, num_sprites=3), max_episode_length=50,
      num_envs=4, device_reset=False)
  env.reset()
  sp = env.sprites(1)[2]
  shape0, angle0, scale0 = sp.shape, sp.angle, sp.scale
  assert (angle0, scale0) == (30.0, 0.15) and shape0 in ('square', 'triangle', 'star_5')
  pos = sp.position
  path = (mpl_transforms.Affine2D().scale(scale0) + mpl_transforms.Affine2D().rotate_deg(angle0)).transform_path(
      mpl_path.Path(shapes.SHAPES[shape0]))

  def vertices(p):
    return mpl_transforms.Affine2D().translate(*pos).transform_path(p).vertices

  assert np.array_equal(sp.vertices, vertices(path))
  sp.angle = 75.0
  path = mpl_transforms.Affine2D().rotate_deg(75.0 - angle0).transform_path(path)
  assert np.array_equal(sp.vertices, vertices(path)) and sp.angle == 75.0
  sp.scale = 0.4
  path = mpl_transforms.Affine2D().scale(0.4 - scale0).transform_path(path)
  assert np.array_equal(sp.vertices, vertices(path)) and sp.scale == 0.4
  sp.shape = 'circle'                      # _reset_centered_path with the CURRENT scale and angle
  path = (mpl_transforms.Affine2D().scale(0.4) + mpl_transforms.Affine2D().rotate_deg(75.0)).transform_path(
      mpl_path.Path(shapes.SHAPES['circle']))
  assert np.array_equal(sp.vertices, vertices(path)) and sp.shape == 'circle'
  assert sp.factors['shape'] == 'circle' and sp.factors['angle'] == 75.0 and sp.factors['scale'] == 0.4
  row = env.observation()['factors'][1, 2].cpu().numpy()
  assert row[2] == shapes.shape_index('circle') + 1 and row[3] == 75.0 and row[4] == 0.4
  # environment 1 now has a target (the circle, far from the goal): its episode goes on; the others, without
  # targets, succeed vacuously (tasks.py:139-142) and end
  ts = env.step(env.null_actions())
  assert ts.step_type.cpu().numpy().tolist() == [2, 1, 2, 2]
  env.close()


def setter_under_a_position_filter_case():
  """Round 6: a filter keyed on an attribute AND on position -- Product([Discrete('shape', ['circle']), Continuous('x', 0, 0.5)]).
  A setter re-labels the sprite in every cell of the filter's position grid (swb_set_sprite_cell_labels), and where the sprite
  stands decides, step by step, whether it counts: the reference's `contains(sprite.factors)` at every step
  (tasks.py:134-137).  The sprite that becomes a circle is a target while it is in the left half, and stops being one when it is
  dragged to the right."""
  from spriteworld_amd import action_spaces, environment, renderers, sprite_generators, tasks
  from spriteworld_amd import factor_distributions as distribs
  np.random.seed(6)
  left = distribs.Product([
      distribs.Continuous('x', 0.1, 0.4), distribs.Continuous('y', 0.3, 0.7), distribs.Discrete('shape', ['square']),
      distribs.Discrete('scale', [0.2]), distribs.Continuous('c0', 0., 1.), distribs.Continuous('c1', 0.5, 1.),
      distribs.Continuous('c2', 0.9, 1.)])
  env = environment.BatchedEnvironment(
      task=tasks.FindGoalPosition(filter_distrib=distribs.Product([distribs.Discrete('shape', ['circle']), distribs.Continuous('x', 0., 0.5)]),
                                  goal_position=(3., 3.), terminate_distance=0.01),
      action_space=action_spaces.DragAndDrop(scale=1.0),
      renderers={'image': renderers.PILRenderer(image_size=(32, 32), anti_aliasing=2, color_to_rgb=renderers.hsv_to_rgb)},
      init_sprites=sprite_generators.generate_sprites(left, num_sprites=1), max_episode_length=50, num_envs=3, device_reset='auto')
  env.reset()
  # no circle anywhere: no sprite passes the filter, the (vacuous) success ends every episode -- except where one becomes a circle
  env.sprites(1)[0].shape = 'circle'
  ts = env.step(env.null_actions())
  assert ts.step_type.cpu().numpy().tolist() == [2, 1, 2]          # environment 1 has a target now (far from the goal)
  assert np.isnan(ts.reward.cpu().numpy()[[0, 2]]).all() and not np.isnan(ts.reward.cpu().numpy()[1])
  # drag that circle into the right half: it is a circle still, but no longer in the filter's x range -> no target, NaN reward,
  # vacuous success, the episode ends
  st = env.state()
  x, y = float(st['x'][1, 0]), float(st['y'][1, 0])
  act = env.null_actions()
  act[1] = torch_row(act, [x, y, min(x + 0.45, 1.0), y])
  env.step(env.null_actions())                                      # (environments 0 and 2 reset; 1 goes on)
  env.sprites(0)[0].shape = 'circle'                                # a circle in environment 0 as well, staying left
  ts = env.step(act)
  r = ts.reward.cpu().numpy()
  assert env.state()['x'][1, 0] >= 0.5
  assert np.isnan(r[1]) and ts.step_type.cpu().numpy()[1] == 2      # moved out of the filter: no target any more
  assert not np.isnan(r[0]) and ts.step_type.cpu().numpy()[0] == 1  # still a target on the left
  env.close()


def torch_row(like, values):
  import torch
  return torch.as_tensor(values, dtype=like.dtype, device=like.device)

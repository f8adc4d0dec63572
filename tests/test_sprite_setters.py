"""Sprite attribute setters on live sprites (reference sprite.py:152-175, pinned by the
reference's tests/sprite_test.py:138-174).

CPU: the oracle's setters against the UNMODIFIED reference (sprites modified between the steps of a running
environment: frames, hit-tests, rewards bit for bit) and the library's host arithmetic against matplotlib itself.
GPU (tests/test_gpu_setters.py): the HIP engine against the oracle.
"""
import copy
import ctypes as C

import numpy as np
import pytest

from oracle import ref_harness
from spriteworld_amd import _abi, _lib, build, shapes

needs_reference = pytest.mark.skipif(not ref_harness.reference_available(), reason='reference tree not present')


def _fresh(episodes):
  yield copy.deepcopy(episodes[0])
  while True:
    for e in episodes:
      yield copy.deepcopy(e)


def _bits(v):
  return np.float64(v).view(np.uint64)


def _script(rng, n_steps, n_sprites_of):
  """Random setter calls between steps: list of (step, sprite, attr, value)."""
  names = list(shapes.SHAPES.keys())
  calls = []
  for t in range(1, n_steps):
    if rng.rand() < 0.35:
      k = int(rng.randint(0, 3))
      value = (str(rng.choice(names)) if k == 0 else
               float(rng.choice([0., 17., 45., 90., 133.5, 270., 359.])) if k == 1 else
               float(rng.choice([0.08, 0.15, 0.2, 0.3, 0.45])))
      calls.append((t, int(rng.randint(0, 64)), ('shape', 'angle', 'scale')[k], value))
  return calls


@needs_reference
@pytest.mark.parametrize('seed', [0, 1, 2, 3])
def test_oracle_setters_equal_reference(seed):
  """A running reference environment whose sprites are modified through the setters (shape, angle, scale -- the
  latter with the reference's (s - old) quirk) and the oracle given the same calls: step types, rewards,
  positions and frames identical; the overrides end at the reset, as fresh sprites replace the old ones."""
  ref_harness.load_reference()
  from spriteworld import action_spaces, environment, renderers, sprite, tasks
  from oracle import oracle
  from spriteworld_amd import lowering
  rng = np.random.RandomState(100 + seed)
  names = list(shapes.SHAPES.keys())

  def gen():
    return [sprite.Sprite(x=float(rng.uniform(0.15, 0.85)), y=float(rng.uniform(0.15, 0.85)),
                          shape=str(rng.choice(names)), angle=float(rng.choice([0, 30, 77, 180, 301])),
                          scale=float(rng.choice([0.1, 0.2, 0.3])), c0=int(rng.randint(0, 256)),
                          c1=int(rng.randint(0, 256)), c2=int(rng.randint(0, 256)))
            for _ in range(int(rng.randint(1, 5)))]

  episodes = [gen() for _ in range(6)]
  S = 4
  task = tasks.FindGoalPosition(goal_position=(0.5, 0.5), terminate_distance=0.05)
  aspace = action_spaces.SelectMove(scale=0.25)
  aa = [1, 2, 5, 3][seed]
  rends = {'image': renderers.PILRenderer(image_size=(32, 32), anti_aliasing=aa)}
  cfg = lowering.lower_config(task, aspace, rends, True, 12, 1, S,
                              pos_is_f32=(lowering.position_dtype(episodes) == np.float32))
  pool = lowering.lower_episodes(episodes, task, rends, max_sprites=S).assign_round_robin(1)
  eng = oracle.Engine(cfg, pool)
  it = _fresh(episodes)
  env = environment.Environment(task=task, action_space=aspace, renderers=rends, init_sprites=lambda: next(it),
                                keep_in_frame=True, max_episode_length=12)
  n_steps = 60
  calls = _script(rng, n_steps, None)
  arng = np.random.RandomState(7 + seed)
  attr_id = {'shape': _abi.ATTR_SHAPE, 'angle': _abi.ATTR_ANGLE, 'scale': _abi.ATTR_SCALE}
  applied = 0
  for t in range(n_steps):
    for (tc, k, attr, value) in calls:
      if tc != t or env._reset_next_step or not env._sprites:
        continue
      k %= len(env._sprites)
      setattr(env._sprites[k], attr, value)
      eng.set_sprite_attr(0, k, attr_id[attr], shapes.shape_index(value) if attr == 'shape' else value)
      applied += 1
      got = eng.get_sprite(0, k)
      assert np.array_equal(got['path'], env._sprites[k]._centered_path.vertices), (t, attr, value)
    # click on a sprite half of the time, so the hit-test runs on the modified paths
    a = arng.uniform(0, 1, 4)
    if env._sprites and arng.rand() < 0.5 and not env._reset_next_step:
      sp = env._sprites[int(arng.randint(0, len(env._sprites)))]
      a[:2] = np.clip(np.asarray(sp.position, dtype=np.float64) + arng.uniform(-0.05, 0.05, 2), 0, 1)
    ts = env.step(a)
    out = eng.step(a[None])
    assert int(ts.step_type) == int(out['step_type'][0]), t
    r = np.nan if ts.reward is None else float(ts.reward)
    assert (np.isnan(r) and np.isnan(out['reward'][0])) or _bits(r) == _bits(out['reward'][0]), (t, r)
    assert np.array_equal(ts.observation['image'], out['obs'][0]), t
    st = eng.state()
    pos = np.array([s.position for s in env._sprites], dtype=np.float64).reshape(-1, 2)
    n = st['n_sprites'][0]
    assert n == len(pos) and np.array_equal(pos[:, 0], st['x'][0, :n]) and np.array_equal(pos[:, 1], st['y'][0, :n]), t
  assert applied >= 5


def _path_op(attr, a, b, verts):
  build.build()
  lib = C.CDLL(_lib.LIB_PATH)
  lib.swb_sprite_path_op.argtypes = [C.c_int32, C.c_double, C.c_double, C.c_int32, C.c_void_p, C.c_void_p]
  v = np.ascontiguousarray(verts, dtype=np.float64)
  out = np.zeros_like(v)
  assert lib.swb_sprite_path_op(attr, float(a), float(b), len(v), v.ctypes.data, out.ctypes.data) == 0
  return out


def test_library_host_arithmetic_equals_matplotlib():
  """swb_sprite_path_op (the arithmetic swb_set_sprite_attr applies, pure host code) against matplotlib's own
  Affine2D / transform_path for every shape and a grid of angles and scales -- bit for bit."""
  from matplotlib import path as mpl_path
  from matplotlib import transforms as mpl_transforms
  rng = np.random.RandomState(3)
  for name, verts in shapes.SHAPES.items():
    for _ in range(6):
      scale, angle = float(rng.uniform(0.05, 0.5)), float(rng.choice([0, 45, 90, 123.25, 200, 359]))
      p = (mpl_transforms.Affine2D().scale(scale) + mpl_transforms.Affine2D().rotate_deg(angle)).transform_path(
          mpl_path.Path(verts))
      fresh = _path_op(_abi.ATTR_SHAPE, scale, angle, verts)
      assert np.array_equal(fresh, p.vertices), name
      a2 = float(rng.uniform(0, 360))
      rot = mpl_transforms.Affine2D().rotate_deg(a2 - angle).transform_path(p)
      assert np.array_equal(_path_op(_abi.ATTR_ANGLE, a2, angle, p.vertices), rot.vertices), name
      s2 = float(rng.uniform(0.05, 0.5))
      sc = mpl_transforms.Affine2D().scale(s2 - scale).transform_path(rot)
      assert np.array_equal(_path_op(_abi.ATTR_SCALE, s2, scale, rot.vertices), sc.vertices), name


def test_reference_setter_known_answers():
  """tests/sprite_test.py:138-174 (testResetShape / testResetAngle / testResetScale) through the library's host
  arithmetic: vertices = centred path + position, to the reference's tolerances."""
  sq = shapes.SHAPES['square']
  fresh = _path_op(_abi.ATTR_SHAPE, 0.25, 0, sq) + 0.5
  assert np.allclose(fresh, [[0.625, 0.625], [0.375, 0.625], [0.375, 0.375], [0.625, 0.375]], atol=1e-3)
  tri = _path_op(_abi.ATTR_SHAPE, 0.25, 0, shapes.SHAPES['triangle']) + 0.5
  assert np.allclose(tri, [[0.5, 0.72], [0.31, 0.39], [0.69, 0.39]], atol=1e-2)
  rot = _path_op(_abi.ATTR_ANGLE, -45, 0, fresh - 0.5) + 0.5
  assert np.allclose(rot, [[0.677, 0.5], [0.5, 0.677], [0.323, 0.5], [0.5, 0.323]], atol=1e-3)
  # the (s - old) quirk: scale 0.25 -> 0.5 multiplies the path by 0.25
  scl = _path_op(_abi.ATTR_SCALE, 0.5, 0.25, fresh - 0.5) + 0.5
  assert np.allclose(scl, [[0.531, 0.531], [0.469, 0.531], [0.469, 0.469], [0.531, 0.469]], atol=1e-3)


# ---------------------------------------------------------------------------------------------------------------
# The GPU scenarios of tests/test_gpu_setters.py, dry-run on CPU against the oracle-backed fake engine: checks the
# scripts themselves and the host-side API (environment.BatchedEnvironment.sprites / set_sprite_attr, LiveSprite).
# ---------------------------------------------------------------------------------------------------------------
def _fake(cfg, pool):
  from tests import _fake_engine
  return _fake_engine.FakeEngine(cfg, pool)


@pytest.mark.parametrize('name,n_envs,steps,aa', [('goal_s5', 6, 5, 5), ('cluster_s5', 6, 4, 1), ('embodied_s12', 3, 3, 5),
                                                   ('ragged_s16', 6, 4, 5), ('geom_256x64', 3, 3, 2), ('geom_96x48', 3, 3, 3),
                                                   ('geom_32x32', 3, 3, 8), ('geom_64x256', 3, 3, 1)])
def test_gpu_scenarios_dry_run(name, n_envs, steps, aa):
  from tests import _setter_cases
  _setter_cases.run_parity(_fake, name, n_envs, steps, aa)


def test_factors_and_reset_scenario_dry_run():
  from tests import _fake_engine, _setter_cases
  _setter_cases.factors_and_reset_case(_fake, _fake_engine.FakeEngineError)


def test_live_sprite_api_on_the_fake_engine(monkeypatch):
  from spriteworld_amd import environment
  from tests import _fake_engine, _setter_cases
  monkeypatch.setattr(environment._engine, 'Engine', _fake_engine.FakeEngine)
  _setter_cases.live_sprite_case()


@pytest.mark.parametrize('backend', ['fake', 'emu'])
def test_setter_under_a_filter_that_also_keys_on_position(monkeypatch, backend):
  from spriteworld_amd import environment
  from tests import _emu_engine, _fake_engine, _setter_cases
  monkeypatch.setattr(environment._engine, 'Engine', _fake_engine.FakeEngine if backend == 'fake' else _emu_engine.EmuTorchEngine)
  _setter_cases.setter_under_a_position_filter_case()


@needs_reference
def test_setters_on_float32_attributes_equal_the_reference_sprite(monkeypatch):
  """factor distributions give sprites np.float32 scales; the reference's setter then takes `s - self._scale`
  in float32 (NEP 50), and a value assigned once keeps the type it was given.  `LiveSprite` on a BatchedEnvironment against
  the UNMODIFIED reference Sprite, vertices bit for bit, through a chain of assignments."""
  from spriteworld_amd import action_spaces, environment, renderers, sprite_generators, tasks
  from spriteworld_amd import factor_distributions as distribs
  from tests import _fake_engine
  ref = ref_harness.load_reference()
  monkeypatch.setattr(environment._engine, 'Engine', _fake_engine.FakeEngine)
  np.random.seed(9)
  factors = distribs.Product([
      distribs.Continuous('x', 0.2, 0.8), distribs.Continuous('y', 0.2, 0.8),
      distribs.Discrete('shape', ['square', 'triangle', 'star_5']), distribs.Continuous('scale', 0.1, 0.25),
      distribs.Discrete('angle', [30.0]), distribs.Continuous('c0', 0., 1.), distribs.Continuous('c1', 0.5, 1.),
      distribs.Continuous('c2', 0.9, 1.)])
  env = environment.BatchedEnvironment(
      task=tasks.NoReward(), action_space=action_spaces.SelectMove(scale=0.0),
      renderers={'image': renderers.PILRenderer(image_size=(64, 64), anti_aliasing=5)},
      init_sprites=sprite_generators.generate_sprites(factors, num_sprites=3), max_episode_length=50,
      num_envs=4, device_reset=False)
  env.reset()
  differs = 0
  for e in range(4):
    for k in range(3):
      sp = env.sprites(e)[k]
      scale0 = sp.scale
      assert float(np.float32(scale0)) == scale0                       # drawn as float32
      pos = sp.position
      r = ref.sprite.Sprite(x=pos[0], y=pos[1], shape=sp.shape, angle=30.0, scale=np.float32(scale0))
      assert np.array_equal(_bits(sp.vertices), _bits(r.vertices))
      for value in (0.3, np.float32(0.17), 0.41):                      # float32 - then a Python float - then a float32 attribute
        differs += float(value - np.float32(r.scale) if isinstance(r.scale, np.float32) else 0.0) != float(value) - float(r.scale)
        sp.scale = value
        r.scale = value
        assert np.array_equal(_bits(sp.vertices), _bits(r.vertices)), (e, k, value)
      sp.angle = 77.5
      r.angle = 77.5
      assert np.array_equal(_bits(sp.vertices), _bits(r.vertices))
  assert differs > 0        # (the float32 subtraction is not the float64 one: the test can tell them apart)
  env.close()

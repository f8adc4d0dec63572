"""dm_env conformance of the N = 1 `Environment`, re-expressing dm_env.test_utils.EnvironmentTestMixin
as the reference uses it (reference: tests/environment_test.py:30-51): reset / step protocol on fresh
environments, and every time step of a longer action sequence conforms to reward_spec(),
discount_spec() and observation_spec() (a dict of specs, as the reference's assertValidObservation
override handles); sampled actions conform to action_spec()."""
import numpy as np
import pytest

from spriteworld_amd import action_spaces, renderers, tasks
from spriteworld_amd import dm_env_compat as dm_env
from spriteworld_amd.sprite import Sprite

pytestmark = pytest.mark.gpu


def _conforms(value, spec):
  """dm_env.test_utils.EnvironmentTestMixin.assertConformsToSpec (spec.validate)."""
  a = np.asarray(value)
  assert a.shape == tuple(spec.shape), (a.shape, spec.shape)
  assert a.dtype == np.dtype(spec.dtype), (a.dtype, spec.dtype)
  if hasattr(spec, 'minimum'):
    assert np.all(a >= spec.minimum) and np.all(a <= spec.maximum), (a, spec.minimum, spec.maximum)


def _valid_step(env, ts):
  assert isinstance(ts, dm_env.TimeStep)
  assert isinstance(ts.step_type, dm_env.StepType)
  if ts.step_type == dm_env.StepType.FIRST:
    assert ts.reward is None and ts.discount is None
  else:
    _conforms(ts.reward, env.reward_spec())
    _conforms(ts.discount, env.discount_spec())
  spec = env.observation_spec()
  assert set(ts.observation) == set(spec)
  for k, v in ts.observation.items():
    _conforms(v, spec[k])


def _reference_test_env():
  """make_object_under_test of tests/environment_test.py:42-51."""
  from spriteworld_amd import environment
  return environment.Environment(task=tasks.NoReward(), action_space=action_spaces.SelectMove(), renderers={},
                                 init_sprites=lambda: [Sprite(c0=255)], max_episode_length=7)


def _rendered_env():
  from spriteworld_amd import environment
  rend = {'image': renderers.PILRenderer(image_size=(64, 64), anti_aliasing=5, color_to_rgb=renderers.hsv_to_rgb),
          'success': renderers.Success()}
  return environment.Environment(task=tasks.FindGoalPosition(terminate_distance=0.2), action_space=action_spaces.SelectMove(scale=0.5),
                                 renderers=rend, init_sprites=lambda: [Sprite(x=0.2, y=0.3, c0=0.3, c1=0.8, c2=0.9),
                                                                       Sprite(x=0.7, y=0.6, shape='circle', c0=0.6, c1=0.9, c2=1.0)],
                                 max_episode_length=9)


@pytest.mark.parametrize('make', [_reference_test_env, _rendered_env])
def test_reset_and_step_protocol_on_fresh_environments(make):
  env = make()                                        # test_reset / test_reset_on_new_env
  ts = env.reset()
  assert ts.first()
  _valid_step(env, ts)
  env.close()
  env = make()                                        # test_step_on_fresh_environment: the first step is a reset
  a = env.action_space.sample()
  ts = env.step(a)
  assert ts.first()
  _valid_step(env, ts)
  ts = env.step(a)                                    # test_step_after_reset
  assert not ts.first()
  _valid_step(env, ts)
  env.close()


@pytest.mark.parametrize('make', [_reference_test_env, _rendered_env])
def test_longer_action_sequence_conforms_to_the_specs(make):
  env = make()
  np.random.seed(5)
  ts = env.reset()
  prev_last = False
  seen = set()
  for _ in range(40):
    a = env.action_space.sample()
    spec = env.action_spec()
    assert np.asarray(a).shape == tuple(spec.shape) and np.asarray(a).dtype.kind == 'f'
    assert np.all(np.asarray(a) >= spec.minimum) and np.all(np.asarray(a) <= spec.maximum)
    ts = env.step(a)
    _valid_step(env, ts)
    assert ts.first() == prev_last                    # auto-reset: FIRST exactly after a LAST step
    prev_last = ts.last()
    seen.add(int(ts.step_type))
  assert seen == {0, 1, 2}
  env.close()


def test_specs_are_specs():
  env = _rendered_env()
  obs = env.observation_spec()
  assert tuple(obs['image'].shape) == (64, 64, 3) and obs['image'].dtype == np.uint8
  assert tuple(obs['success'].shape) == () and obs['success'].dtype == np.bool_
  assert tuple(env.reward_spec().shape) == () and np.dtype(env.reward_spec().dtype).kind == 'f'
  d = env.discount_spec()
  assert float(d.minimum) == 0.0 and float(d.maximum) == 1.0
  env.close()

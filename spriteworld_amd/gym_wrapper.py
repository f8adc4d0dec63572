"""Gym-style surface over the engine (reference gym_wrapper.py:26-135).

`GymWrapper(env)` wraps the N = 1 `environment.Environment` exactly like the reference wrapper:
`step -> (obs, reward or 0, done, {'discount': ...})`, boolean observations as float32, `render()`
returning the last image.  `BatchedGymWrapper(env)` is the same contract over a
`BatchedEnvironment`: every value is a device tensor with a leading N axis and the rewards /
discounts of FIRST steps (None in the reference) are 0 / NaN.

`gym` is optional: without it the spaces are the small stand-ins below (same attributes).
"""
import numpy as np
import torch

from spriteworld_amd import dm_env_compat as dm_env

specs = dm_env.specs

try:  # pragma: no cover - gym is not installed in the build image
  from gym import spaces
except ImportError:

  class _Space(object):

    def __repr__(self):
      return '%s(%s)' % (type(self).__name__, ', '.join('%s=%r' % kv for kv in sorted(vars(self).items())))

  class _Spaces(object):

    class Box(_Space):

      def __init__(self, low, high, shape=None, dtype=np.float32):
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape or ()), np.dtype(dtype)

      def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

      def sample(self):
        return np.random.uniform(self.low, self.high, size=self.shape).astype(self.dtype)

      def __eq__(self, other):
        return type(other) is type(self) and vars(other) == vars(self)

    class Discrete(_Space):

      def __init__(self, n):
        self.n = int(n)

      def contains(self, x):
        return 0 <= int(x) < self.n

      def sample(self):
        return int(np.random.randint(self.n))

      def __eq__(self, other):
        return type(other) is type(self) and other.n == self.n

    class Tuple(_Space):

      def __init__(self, spaces_):
        self.spaces = tuple(spaces_)

      def sample(self):
        return tuple(s.sample() for s in self.spaces)

      def __eq__(self, other):
        return type(other) is type(self) and other.spaces == self.spaces

    class Dict(_Space):

      def __init__(self, spaces_):
        self.spaces = dict(spaces_)

      def sample(self):
        return {k: s.sample() for k, s in self.spaces.items()}

      def __eq__(self, other):
        return type(other) is type(self) and other.spaces == self.spaces

  spaces = _Spaces


def _spec_to_space(spec):
  """dm_env.specs -> gym.spaces (gym_wrapper.py:26-39)."""
  if isinstance(spec, list):
    return spaces.Tuple([_spec_to_space(s) for s in spec])
  if isinstance(spec, specs.DiscreteArray):
    return spaces.Discrete(spec.num_values)
  if isinstance(spec, specs.BoundedArray):
    return spaces.Box(np.asarray(spec.minimum).item(), np.asarray(spec.maximum).item(), shape=spec.shape,
                      dtype=spec.dtype)
  raise ValueError('Unknown type for specs: {}'.format(spec))


class _GymSurface(object):
  """What both wrappers share: lazily built spaces, attribute passthrough, the cached last frame."""
  metadata = {'render.modes': ['rgb_array']}
  _leading = ()                      # leading axes of every observation (the batch axis, if any)

  def __init__(self, env):
    self._env = env
    self._frame = None
    self._spaces = {}
    self._env.reset()                # the reference resets on construction (gym_wrapper.py:58)

  def __getattr__(self, name):
    return getattr(self._env, name)

  @property
  def observation_space(self):
    if 'obs' not in self._spaces:
      self._spaces['obs'] = spaces.Dict({
          key: spaces.Box(-np.inf, np.inf, self._leading + tuple(spec.shape), dtype=spec.dtype)
          for key, spec in self._env.observation_spec().items()})
    return self._spaces['obs']

  @property
  def action_space(self):
    if 'act' not in self._spaces:
      self._spaces['act'] = _spec_to_space(self._env.action_spec())
    return self._spaces['act']

  def reset(self):
    return self._observations(self._env.reset().observation)

  def render(self, mode='rgb_array'):
    del mode                         # rendering always happens; this returns the last 'image'
    return self._frame


class GymWrapper(_GymSurface):
  """Gym interface of a single environment (`environment.Environment`): numpy observations keyed
  like the `renderers` dict, booleans as float32, `reward or 0`, `done = last()`."""

  def _observations(self, obs):
    out = {}
    for key, value in obs.items():
      value = np.asarray(value)
      out[key] = value.astype(np.float32) if value.dtype == np.bool_ else value
      if key == 'image':
        self._frame = out[key]
    return out

  def step(self, action):
    ts = self._env.step(action)
    return self._observations(ts.observation), (ts.reward or 0), ts.last(), {'discount': ts.discount}

  def close(self):
    pass


class BatchedGymWrapper(_GymSurface):
  """The same contract for N environments: tensors with a leading N axis, auto-reset included.

  An environment whose step returned done=True restarts on the next `step` (its action is ignored,
  like the reference's Environment, environment.py:90-91) and reports reward 0 on that FIRST step.
  """

  def __init__(self, env):
    self._leading = (env.num_envs,)
    super(BatchedGymWrapper, self).__init__(env)

  def _observations(self, obs):
    out = {}
    for key, value in obs.items():
      out[key] = value.to(torch.float32) if value.dtype == torch.bool else value
      if key == 'image':
        self._frame = out[key]
    return out

  def step(self, actions):
    ts = self._env.step(actions)
    reward = torch.nan_to_num(ts.reward, nan=0.0)           # `reward or 0`
    done = ts.step_type == int(dm_env.StepType.LAST)
    return self._observations(ts.observation), reward, done, {'discount': ts.discount}

  def close(self):
    self._env.close()

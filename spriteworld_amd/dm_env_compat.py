"""dm_env types used at the environment boundary.

The reference returns `dm_env.TimeStep`s (reference: spriteworld/environment.py:
78,105-108) and `dm_env.specs` arrays.  When the real `dm_env` package is
installed it is used; otherwise this module provides structurally identical
stand-ins (StepType, TimeStep with first()/mid()/last(), restart/transition/
termination, specs.Array/BoundedArray/DiscreteArray).
"""
import collections
import enum

import numpy as np

try:  # pragma: no cover - dm_env is not installed in the build image
  from dm_env import StepType, TimeStep, restart, transition, termination, specs  # noqa: F401
  HAVE_DM_ENV = True
except ImportError:
  HAVE_DM_ENV = False

  class StepType(enum.IntEnum):
    FIRST = 0
    MID = 1
    LAST = 2

    def first(self):
      return self is StepType.FIRST

    def mid(self):
      return self is StepType.MID

    def last(self):
      return self is StepType.LAST

  class TimeStep(collections.namedtuple('TimeStep',
                                        ['step_type', 'reward', 'discount', 'observation'])):
    __slots__ = ()

    def first(self):
      return self.step_type == StepType.FIRST

    def mid(self):
      return self.step_type == StepType.MID

    def last(self):
      return self.step_type == StepType.LAST

  def restart(observation):
    return TimeStep(StepType.FIRST, None, None, observation)

  def transition(reward, observation, discount=1.0):
    return TimeStep(StepType.MID, reward, discount, observation)

  def termination(reward, observation):
    return TimeStep(StepType.LAST, reward, 0.0, observation)

  class _Specs(object):

    class Array(object):

      def __init__(self, shape, dtype, name=None):
        self.shape = tuple(int(d) for d in shape)
        self.dtype = np.dtype(dtype)
        self.name = name

      def __repr__(self):
        return 'Array(shape={}, dtype={}, name={})'.format(self.shape, self.dtype, self.name)

  class _BoundedArray(_Specs.Array):

    def __init__(self, shape, dtype, minimum, maximum, name=None):
      super(_BoundedArray, self).__init__(shape, dtype, name)
      self.minimum = np.asarray(minimum)
      self.maximum = np.asarray(maximum)

  class _DiscreteArray(_BoundedArray):

    def __init__(self, num_values, dtype=np.int32, name=None):
      super(_DiscreteArray, self).__init__((), dtype, 0, num_values - 1, name)
      self.num_values = num_values

  _Specs.BoundedArray = _BoundedArray
  _Specs.DiscreteArray = _DiscreteArray
  specs = _Specs

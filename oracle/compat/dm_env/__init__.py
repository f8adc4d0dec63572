"""Minimal stand-in for the `dm_env` package (absent from this image).

TEST INFRASTRUCTURE ONLY.  It exists so that the *unmodified* reference
(`/root/reference/spriteworld`) can be imported by `oracle/ref_harness.py`
to pin the oracle and to generate golden vectors.  It restates the public
dm_env API surface the reference touches (reference call sites:
spriteworld/environment.py:22,27,78,105-108; spriteworld/action_spaces.py:25,
62-63,161-164; spriteworld/renderers/pil_renderer.py:22,61-62):
StepType, TimeStep, Environment, restart/transition/termination, specs.
"""
import abc
import collections
import enum

from . import specs  # noqa: F401


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


class TimeStep(
    collections.namedtuple('TimeStep',
                           ['step_type', 'reward', 'discount', 'observation'])):
  __slots__ = ()

  def first(self):
    return self.step_type == StepType.FIRST

  def mid(self):
    return self.step_type == StepType.MID

  def last(self):
    return self.step_type == StepType.LAST


class Environment(metaclass=abc.ABCMeta):

  @abc.abstractmethod
  def reset(self):
    pass

  @abc.abstractmethod
  def step(self, action):
    pass

  @abc.abstractmethod
  def observation_spec(self):
    pass

  @abc.abstractmethod
  def action_spec(self):
    pass

  def close(self):
    pass


def restart(observation):
  return TimeStep(StepType.FIRST, None, None, observation)


def transition(reward, observation, discount=1.0):
  return TimeStep(StepType.MID, reward, discount, observation)


def termination(reward, observation):
  return TimeStep(StepType.LAST, reward, 0.0, observation)

"""Action-space descriptors.

Mirrors of the reference constructors and attribute names (reference:
spriteworld/action_spaces.py:29-111 SelectMove, :114-137 DragAndDrop, :140-221
Embodied).  `step()` is not implemented on the host: applying an action is part
of the HIP cover kernel; these objects only describe *which* action space
the batch uses (`lowering.lower_config` reads `_scale`, `_motion_cost`,
`_noise_scale`, `_step_size`, exactly the attributes the reference sets) and
provide `sample()` / `action_spec()`.
"""
import numpy as np

from spriteworld_amd import dm_env_compat as dm_env

specs = dm_env.specs


class SelectMove(object):
  """Two clicks: [select_x, select_y, move_x, move_y] in [0, 1]^4."""

  def __init__(self, scale=1.0, motion_cost=0.0, noise_scale=None):
    self._scale = scale
    self._motion_cost = motion_cost
    self._noise_scale = noise_scale
    self._action_spec = specs.BoundedArray(shape=(4,), dtype=np.float32, minimum=0.0, maximum=1.0)

  def sample(self, num_envs=None):
    """Uniform random action(s); float64 like the reference's `sample()`."""
    if num_envs is None:
      return np.random.uniform(0., 1., size=(4,))
    return np.random.uniform(0., 1., size=(num_envs, 4))

  def action_spec(self):
    return self._action_spec


class DragAndDrop(SelectMove):
  """As SelectMove, but the motion is scale * (second click - first click)."""


class Embodied(object):
  """sprites[-1] is the agent body: action = (carry in {0,1}, direction in {0..3})."""

  def __init__(self, step_size=0.05, motion_cost=0.):
    self._step_size = step_size
    self._motion_cost = motion_cost
    self._action_spec = [
        specs.DiscreteArray(num_values=2, dtype=np.int64),
        specs.DiscreteArray(num_values=4, dtype=np.int64),
    ]
    s = self._step_size
    self.action_to_motion = {
        0: np.array([0, s]),   # up
        1: np.array([-s, 0]),  # left
        2: np.array([0, -s]),  # down
        3: np.array([s, 0]),   # right
    }

  def sample(self, num_envs=None):
    if num_envs is None:
      return [np.random.randint(0, 2), np.random.randint(0, 4)]
    return np.stack([np.random.randint(0, 2, size=num_envs),
                     np.random.randint(0, 4, size=num_envs)], axis=1)

  def action_spec(self):
    return self._action_spec

"""Task descriptors.

Mirrors of the reference constructors and attribute names (reference:
spriteworld/tasks.py:70-81 NoReward, :84-158 FindGoalPosition, :161-245
Clustering, :248-296 MetaAggregated).  Reward and success are evaluated by the
HIP step kernel for all environments at once; these objects carry the
parameters (`lowering.lower_config` reads the same private attributes the
reference sets) and the per-sprite membership rule (`filter_distrib`,
`cluster_distribs`), which is static per episode and evaluated on the host when
an episode pool is built.
"""
import numpy as np


class NoReward(object):
  """No task: reward 0.0, never succeeds."""


class FindGoalPosition(object):
  """Bring every sprite passing `filter_distrib` within `terminate_distance` of the goal."""

  def __init__(self, filter_distrib=None, goal_position=(0.5, 0.5), terminate_distance=0.05,
               terminate_bonus=0.0, weights_dimensions=(1, 1), sparse_reward=False,
               raw_reward_multiplier=50):
    self._filter_distrib = filter_distrib
    self._goal_position = np.asarray(goal_position)
    self._terminate_bonus = terminate_bonus
    self._terminate_distance = terminate_distance
    self._sparse_reward = sparse_reward
    self._weights_dimensions = np.asarray(weights_dimensions)
    self._raw_reward_multiplier = raw_reward_multiplier


class Clustering(object):
  """Reward 1 / Davies-Bouldin of the sprite positions grouped by `cluster_distribs`."""

  def __init__(self, cluster_distribs, termination_threshold=2.5, terminate_bonus=0.0,
               sparse_reward=False, reward_range=10):
    self._cluster_distribs = cluster_distribs
    self._num_clusters = len(cluster_distribs)
    self._termination_threshold = termination_threshold
    self._terminate_bonus = terminate_bonus
    self._sparse_reward = sparse_reward
    self._reward_range = reward_range


class MetaAggregated(object):
  """nan-aware sum/max/min/mean of sub-task rewards; all/any of their successes."""
  REWARD_AGGREGATOR = {'sum': np.nansum, 'max': np.nanmax, 'min': np.nanmin, 'mean': np.nanmean}
  TERMINATION_CRITERION = {'all': np.all, 'any': np.any}

  def __init__(self, subtasks, reward_aggregator='sum', termination_criterion='all',
               terminate_bonus=0.0):
    if reward_aggregator not in MetaAggregated.REWARD_AGGREGATOR:
      raise ValueError('Unknown reward_aggregator. {} not in {}'.format(
          reward_aggregator, MetaAggregated.REWARD_AGGREGATOR))
    if termination_criterion not in MetaAggregated.TERMINATION_CRITERION:
      raise ValueError('Unknown termination_criterion. {} not in {}'.format(
          termination_criterion, MetaAggregated.TERMINATION_CRITERION))
    self._subtasks = subtasks
    self._reward_aggregator = MetaAggregated.REWARD_AGGREGATOR[reward_aggregator]
    self._termination_criterion = MetaAggregated.TERMINATION_CRITERION[termination_criterion]
    self._terminate_bonus = terminate_bonus

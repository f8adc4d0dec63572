"""Tasks whose filters / cluster distributions key on POSITION (round 6): the reference evaluates `contains(sprite.factors)` at
every step (tasks.py:134-137, 196-205), so a sprite's membership changes as it moves.  One definition of the cases, written
against a namespace `ns` with the attributes `tasks`, `distribs`, `Sprite`, `action_spaces`, `renderers` -- the REFERENCE's
modules (oracle pinned against the reference, golden fixtures generated from it) or this package's mirrors (kernel against the
oracle where the reference is absent).  TEST INFRASTRUCTURE ONLY."""
import numpy as np

IMAGE, AA = 32, 3
N_SPRITES = 6


def namespace_of_mirrors():
  import types
  from spriteworld_amd import action_spaces, factor_distributions, renderers, sprite, tasks
  return types.SimpleNamespace(tasks=tasks, distribs=factor_distributions, Sprite=sprite.Sprite, action_spaces=action_spaces,
                               renderers=renderers)


def namespace_of_reference():
  import types
  from spriteworld import action_spaces, factor_distributions, renderers, sprite, tasks
  return types.SimpleNamespace(tasks=tasks, distribs=factor_distributions, Sprite=sprite.Sprite, action_spaces=action_spaces,
                               renderers=renderers)


CASES = ('goal_x_lt_half', 'goal_c0_and_yband', 'goal_two_xbands', 'goal_all_but_a_corner', 'goal_whole_frame_half_open',
         'goal_f64_bounds', 'meta_swap_sides', 'cluster_by_side')


def task_of(ns, name):
  D, T = ns.distribs, ns.tasks
  if name == 'goal_x_lt_half':
    return T.FindGoalPosition(filter_distrib=D.Continuous('x', 0., 0.5), goal_position=(0.75, 0.5), terminate_distance=0.2)
  if name == 'goal_c0_and_yband':
    return T.FindGoalPosition(filter_distrib=D.Product([D.Continuous('c0', 0., 0.5), D.Continuous('y', 0.25, 0.75)]),
                              goal_position=(0.5, 0.9), terminate_distance=0.15, terminate_bonus=2.)
  if name == 'goal_two_xbands':        # four thresholds on one axis
    return T.FindGoalPosition(filter_distrib=D.Mixture([D.Continuous('x', 0., 0.3), D.Continuous('x', 0.6, 0.9)]),
                              goal_position=(0.45, 0.5), terminate_distance=0.12, sparse_reward=True)
  if name == 'goal_all_but_a_corner':  # everything except the upper right quadrant
    whole = D.Product([D.Continuous('x', 0., 1.5), D.Continuous('y', 0., 1.5)])
    corner = D.Product([D.Continuous('x', 0.5, 1.5), D.Continuous('y', 0.5, 1.5)])
    return T.FindGoalPosition(filter_distrib=D.SetMinus(whole, corner), goal_position=(0.8, 0.8), terminate_distance=0.25)
  if name == 'goal_whole_frame_half_open':   # [0, 1): a sprite clipped to x = 1.0 drops out, one clipped to x = 0.0 stays in
    return T.FindGoalPosition(filter_distrib=D.Continuous('x', 0., 1.), goal_position=(1.0, 0.5), terminate_distance=0.1)
  if name == 'goal_f64_bounds':        # np.float64 bounds: against float32 positions numpy compares in float64
    return T.FindGoalPosition(filter_distrib=D.Continuous('x', np.float64(0.3), np.float64(0.7)), goal_position=(0.1, 0.1),
                              terminate_distance=0.2)
  if name == 'meta_swap_sides':
    left = T.FindGoalPosition(filter_distrib=D.Continuous('x', 0., 0.5), goal_position=(0.75, 0.5), terminate_distance=0.3)
    right = T.FindGoalPosition(filter_distrib=D.Continuous('x', 0.5, 1.5), goal_position=(0.25, 0.5), terminate_distance=0.3)
    return T.MetaAggregated([left, right], reward_aggregator='sum', termination_criterion='any', terminate_bonus=1.)
  if name == 'cluster_by_side':        # cluster membership by where a sprite stands (it changes as sprites are moved)
    return T.Clustering([D.Product([D.Continuous('x', 0., 0.5), D.Continuous('y', 0., 1.5)]), D.Continuous('x', 0.5, 1.5)],
                        termination_threshold=1.5, terminate_bonus=1., reward_range=4.)
  raise ValueError(name)


def episodes_of(ns, name, f32, n_episodes=10, seed=0):
  """[[Sprite] * N_SPRITES] * n_episodes; float32 positions (what factor distributions draw) or Python floats (float64)."""
  rng = np.random.RandomState(1000 + seed + 17 * CASES.index(name))
  num = (lambda v: np.float32(v)) if f32 else float

  def one():
    return ns.Sprite(x=num(rng.uniform(0.05, 0.95)), y=num(rng.uniform(0.05, 0.95)),
                     shape=str(rng.choice(['square', 'triangle', 'star_5', 'circle'])), angle=int(rng.randint(0, 360)),
                     scale=float(rng.choice([0.12, 0.2])), c0=num(rng.uniform(0., 1.)), c1=num(0.9), c2=num(1.0))
  return [[one() for _ in range(N_SPRITES)] for _ in range(n_episodes)]


def environment_parts(ns, name):
  """(task, action_space, renderers, keep_in_frame, max_episode_length)."""
  task = task_of(ns, name)
  aspace = ns.action_spaces.SelectMove(scale=0.6, motion_cost=0.1)       # long moves: sprites cross the cuts and hit the frame
  rends = {'image': ns.renderers.PILRenderer(image_size=(IMAGE, IMAGE), anti_aliasing=AA,
                                             color_to_rgb=getattr(ns.renderers, 'hsv_to_rgb', None) or
                                             ns.renderers.color_maps.hsv_to_rgb)}
  return task, aspace, rends, True, 12

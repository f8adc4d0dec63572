"""The reference's own known-answer tests, re-expressed against this engine.

Each case cites the reference test it restates (file:line under /root/reference/tests).  The
reference tests cannot run here as written (absl / mock / dm_env are absent),
so the vectors are fed through the batched boundary instead: sprites -> pool, task/action space ->
SwbConfig, then `step()` on the CPU oracle (`kind='oracle'`), the HIP engine (`kind='hip'`, GPU) or the kernel source
run by the host emulator (`kind='emu'`, tests/emu).
"""
import numpy as np
import pytest

from spriteworld_amd import action_spaces, lowering, renderers, tasks
from spriteworld_amd import factor_distributions as distribs
from spriteworld_amd.sprite import Sprite

# 'emu': the kernel source executed lane by lane on the host (tests/emu, test infrastructure) -- what the CPU suite can
# say about the device code where no GPU exists
KINDS = ['oracle', 'emu', pytest.param('hip', marks=pytest.mark.gpu)]


class Harness(object):
  """One environment holding `sprites`; steps through the oracle or the HIP engine."""

  def __init__(self, kind, sprites, task=None, action_space=None, rends=None, keep_in_frame=False,
               max_episode_length=1000):
    task = task or tasks.NoReward()
    action_space = action_space or action_spaces.SelectMove()
    rends = rends if rends is not None else {}
    episodes = [list(sprites)]
    S = max(len(sprites), 1)
    pos_dt = lowering.position_dtype(episodes)
    self.cfg = lowering.lower_config(task, action_space, rends, keep_in_frame, max_episode_length, 1, S,
                                     pos_is_f32=(pos_dt == np.float32))
    self.pool = lowering.lower_episodes(episodes, task, rends, max_sprites=S).assign_round_robin(1)
    self.kind = kind
    if kind == 'oracle':
      from oracle import oracle
      self.eng = oracle.Engine(self.cfg, self.pool)
    elif kind == 'emu':
      from tests import _emu_engine
      self.eng = _emu_engine.EmuEngine(self.cfg, self.pool)
    else:
      from spriteworld_amd import engine
      self.eng = engine.Engine(self.cfg, self.pool)
    self.has_image = lowering.find_pil_renderer(rends)[1] is not None

  def step(self, action):
    a = np.asarray(action)[None]
    if self.kind == 'oracle':
      out = self.eng.step(a, render=self.has_image)
    else:
      self.eng.step(a, render=self.has_image)
      out = self.eng.outputs_host()
    st = self.eng.state()
    n = st['n_sprites'][0]
    return dict(step_type=int(out['step_type'][0]), reward=float(out['reward'][0]),
                success=bool(out['success'][0]), obs=out['obs'][0] if self.has_image else None,
                pos=np.stack([st['x'][0, :n], st['y'][0, :n]], 1))

  def set_positions(self, pos):
    pos = np.asarray(pos, dtype=np.float64)
    S = self.cfg.max_sprites
    x, y = np.zeros((1, S)), np.zeros((1, S))
    x[0, :len(pos)], y[0, :len(pos)] = pos[:, 0], pos[:, 1]
    self.eng.set_positions(x, y)


NOOP = np.array([0.999, 0.001, 0.5, 0.5])   # clicks empty space, zero motion


def _task_eval(kind, sprites, task):
  h = Harness(kind, sprites, task=task)
  assert h.step(NOOP)['step_type'] == 0
  out = h.step(NOOP)
  return out['reward'], out['success']


# --------------------------------------------------------------------------- action spaces
@pytest.mark.parametrize('kind', KINDS)
def test_select_move_script(kind):
  """tests/action_spaces_test.py:53-98 SelectMoveTest.testMoveSprites."""
  sprites = [Sprite(x=0.55, y=0.5), Sprite(x=0.5, y=0.5)]
  script = [  # action, keep_in_frame, expected positions
      ([0.52, 0.52, 0.5, 0.48], False, [[0.55, 0.5], [0.5, 0.49]]),
      ([0.58, 0.5, 0.9, 0.9], False, [[0.75, 0.7], [0.5, 0.49]]),
      ([0.58, 0.5, 0.9, 0.9], False, [[0.75, 0.7], [0.5, 0.49]]),
      ([0.5, 0.5, 0.2, 0.5], False, [[0.75, 0.7], [0.35, 0.49]]),
      ([0.78, 0.74, 0.9, 0.9], False, [[0.95, 0.9], [0.35, 0.49]]),
      ([0.92, 0.9, 0.9, 0.5], True, [[1., 0.9], [0.35, 0.49]]),
      ([0.98, 0.9, 0.7, 0.9], False, [[1.1, 1.1], [0.35, 0.49]]),
  ]
  pos = np.array([[0.55, 0.5], [0.5, 0.49 + 0.01]])
  for action, keep, want in script:
    h = Harness(kind, sprites, action_space=action_spaces.SelectMove(scale=0.5), keep_in_frame=keep)
    h.step(NOOP)
    h.set_positions(pos)
    pos = h.step(np.array(action))['pos']
    assert np.allclose(pos, want, atol=1e-5), (action, pos)


@pytest.mark.parametrize('kind', KINDS)
def test_drag_and_drop_script(kind):
  """tests/action_spaces_test.py:124-168 DragAndDropTest.testMoveSprites."""
  sprites = [Sprite(x=0.55, y=0.5), Sprite(x=0.5, y=0.5)]
  script = [
      ([0.52, 0.52, 0.52, 0.5], False, [[0.55, 0.5], [0.5, 0.49]]),
      ([0.58, 0.5, 0.98, 0.9], False, [[0.75, 0.7], [0.5, 0.49]]),
      ([0.58, 0.5, 0.9, 0.9], False, [[0.75, 0.7], [0.5, 0.49]]),
      ([0.5, 0.5, 0.2, 0.5], False, [[0.75, 0.7], [0.35, 0.49]]),
      ([0.78, 0.74, 0.98, 0.94], False, [[0.85, 0.8], [0.35, 0.49]]),
      ([0.82, 0.8, 1.3, 1.0], True, [[1., 0.9], [0.35, 0.49]]),
      ([0.99, 0.9, 1.19, 1.3], False, [[1.1, 1.1], [0.35, 0.49]]),
  ]
  pos = np.array([[0.55, 0.5], [0.5, 0.5]])
  for action, keep, want in script:
    h = Harness(kind, sprites, action_space=action_spaces.DragAndDrop(scale=0.5), keep_in_frame=keep)
    h.step(NOOP)
    h.set_positions(pos)
    pos = h.step(np.array(action))['pos']
    assert np.allclose(pos, want, atol=1e-5), (action, pos)


@pytest.mark.parametrize('kind', KINDS)
@pytest.mark.parametrize('scale,action,motion_cost,true_cost', [
    (1, [0.5, 0.5, 0.2, 0.75], 0., 0.), (1, [0.5, 0.5, 0.2, 0.75], 1., -0.39),
    (1, [0.2, 0.3, 0.2, 0.75], 1., -0.39), (0.5, [0.5, 0.5, 0.2, 0.75], 1., -0.195)])
def test_select_move_motion_cost(kind, scale, action, motion_cost, true_cost):
  """tests/action_spaces_test.py:41-51 testMotionCost (sprites=[]: an environment with no sprite)."""
  h = Harness(kind, [], action_space=action_spaces.SelectMove(scale=scale, motion_cost=motion_cost))
  h.step(NOOP)
  assert abs(h.step(np.array(action))['reward'] - true_cost) < 0.01


@pytest.mark.parametrize('kind', KINDS)
@pytest.mark.parametrize('init,action,final,keep', [
    ([[0.5, 0.5], [0.2, 0.8]], (0, 0), [[0.5, 0.5], [0.2, 0.9]], True),
    ([[0.5, 0.5], [0.2, 0.8]], (1, 0), [[0.5, 0.5], [0.2, 0.9]], True),
    ([[0.5, 0.5], [0.45, 0.55]], (1, 3), [[0.6, 0.5], [0.55, 0.55]], True),
    ([[0.5, 0.5], [0.45, 0.55]], (1, 1), [[0.4, 0.5], [0.35, 0.55]], True),
    ([[0.5, 0.5], [0.45, 0.55]], (1, 2), [[0.5, 0.4], [0.45, 0.45]], True),
    ([[0.95, 0.02], [0.95, 0.05]], (1, 3), [[1., 0.02], [1., 0.05]], True),
    ([[0.95, 0.02], [0.95, 0.05]], (1, 3), [[1.05, 0.02], [1.05, 0.05]], False),
    ([[0.45, 0.55], [0.5, 0.5], [0.45, 0.55]], (1, 3), [[0.45, 0.55], [0.6, 0.5], [0.55, 0.55]], True),
])
def test_embodied_scenarios(kind, init, action, final, keep):
  """tests/action_spaces_test.py:185-241 EmbodiedTest.testMoveSprites."""
  sprites = [Sprite(x=p[0], y=p[1], shape='square', scale=0.15) for p in init]
  h = Harness(kind, sprites, action_space=action_spaces.Embodied(step_size=0.1), keep_in_frame=keep)
  h.step(np.array([0, 0]))
  assert np.allclose(h.step(np.array(action))['pos'], final, atol=1e-5)


# --------------------------------------------------------------------------- tasks
def _sprites_at(positions, c0=0):
  return [Sprite(x=float(p[0]), y=float(p[1]), c0=c0) for p in positions]


@pytest.mark.parametrize('kind', KINDS)
@pytest.mark.parametrize('positions,reward,success', [
    ([[0., 0.]], -30.4, False), ([[0.4, 0.6]], -2.1, False), ([[0.43, 0.56]], 0.4, True),
    ([[0.48, 0.52], [0.4, 0.6]], 1.5, False), ([[0.48, 0.52], [0.5, 0.5]], 8.6, True)])
def test_find_goal_basic_reward(kind, positions, reward, success):
  """tests/tasks_test.py:42-54 GoalPositionTest.testBasicReward."""
  r, s = _task_eval(kind, _sprites_at(positions), tasks.FindGoalPosition(goal_position=(0.5, 0.5), terminate_distance=0.1))
  assert abs(r - reward) < 0.1 and s == success


@pytest.mark.parametrize('kind', KINDS)
@pytest.mark.parametrize('positions,dist,reward,success', [
    ([[0.4, 0.6]], 0.15, 0.4, True), ([[0.36, 0.5]], 0.15, 0.5, True), ([[0.34, 0.5]], 0.15, -0.5, False),
    ([[0.34, 0.5]], 0.2, 2., True), ([[0.34, 0.39]], 0.2, 0.2, True), ([[0.34, 0.37]], 0.2, -0.3, False)])
def test_find_goal_terminate_distance(kind, positions, dist, reward, success):
  """tests/tasks_test.py:56-70 testTerminateDistance."""
  r, s = _task_eval(kind, _sprites_at(positions), tasks.FindGoalPosition(goal_position=(0.5, 0.5), terminate_distance=dist))
  assert abs(r - reward) < 0.1 and s == success


@pytest.mark.parametrize('kind', KINDS)
@pytest.mark.parametrize('positions,kwargs,reward', [
    # testTerminateBonus :72-84
    ([[0.4, 0.52]], dict(terminate_bonus=3.), -0.1), ([[0.43, 0.52]], dict(terminate_bonus=3.), 4.4),
    ([[0.43, 0.52]], dict(terminate_bonus=1.), 2.4), ([[0.43, 0.52], [0.4, 0.52]], dict(terminate_bonus=3.), 1.3),
    ([[0.43, 0.52], [0.43, 0.52]], dict(terminate_bonus=3.), 5.7),
    # testWeightsDimensions :86-101
    ([[0.43, 0.52]], dict(weights_dimensions=(1, 1)), 1.4), ([[0.43, 0.52]], dict(weights_dimensions=(3, 1)), -1.1),
    ([[0.3, 0.52]], dict(weights_dimensions=(7, 2)), -21.5), ([[0.3, 0.52]], dict(weights_dimensions=(0.1, 0.2)), 1.8),
    # testRewardMultiplier :103-118
    ([[0.35, 0.52]], dict(raw_reward_multiplier=50.0), -2.6), ([[0.35, 0.52]], dict(raw_reward_multiplier=10.0), -0.5),
    ([[0.43, 0.52]], dict(terminate_bonus=1., raw_reward_multiplier=10.0), 1.3),
    ([[0.43, 0.52], [0.4, 0.52]], dict(raw_reward_multiplier=50.0), 1.3),
    ([[0.43, 0.52], [0.43, 0.52]], dict(raw_reward_multiplier=10.0), 0.5),
    # testSparseReward :120-134
    ([[0.35, 0.52]], dict(sparse_reward=True, terminate_bonus=1.), 0.),
    ([[0.43, 0.52]], dict(sparse_reward=True, terminate_bonus=1.), 2.4),
    ([[0.43, 0.52]], dict(sparse_reward=True, terminate_bonus=3.), 4.4),
    ([[0.43, 0.52], [0.4, 0.55]], dict(sparse_reward=True, terminate_bonus=1.), 0.),
    ([[0.43, 0.52], [0.43, 0.52]], dict(sparse_reward=True, terminate_bonus=1.), 3.7),
    ([[0.43, 0.52], [0.43, 0.52]], dict(sparse_reward=True, terminate_bonus=3.), 5.7),
])
def test_find_goal_parameter_tables(kind, positions, kwargs, reward):
  r, _ = _task_eval(kind, _sprites_at(positions),
                    tasks.FindGoalPosition(goal_position=(0.5, 0.5), terminate_distance=0.1, **kwargs))
  assert abs(r - reward) < 0.1


@pytest.mark.parametrize('kind', KINDS)
def test_find_goal_filter_distrib_and_nan(kind):
  """tests/tasks_test.py:136-168 testFilterDistrib / testNoFilteredSprites."""
  sprites = [Sprite(x=0.45, y=0.45, c0=64), Sprite(x=0.45, y=0.55, c0=128), Sprite(x=0.55, y=0.45, c0=192),
             Sprite(x=0.4, y=0.4, c0=255)]
  filters = [distribs.Continuous('c0', 0, 65), distribs.Continuous('c0', 0, 129), distribs.Continuous('c0', 0, 193),
             distribs.Continuous('c0', 0, 256), distribs.Continuous('c0', 65, 256)]
  for f, want_r, want_s in zip(filters, [1.5, 2.9, 4.4, 2.3, 0.9], [True, True, True, False, False]):
    r, s = _task_eval(kind, sprites, tasks.FindGoalPosition(filter_distrib=f, goal_position=(0.5, 0.5), terminate_distance=0.1))
    assert abs(r - want_r) < 0.1 and s == want_s
  r, _ = _task_eval(kind, [Sprite(x=0.45, y=0.45, c0=255)],
                    tasks.FindGoalPosition(filter_distrib=distribs.Continuous('c0', 0, 254), goal_position=(0.5, 0.5),
                                           terminate_distance=0.1))
  assert np.isnan(r)


CLUSTERS = [distribs.Continuous('c0', 0, 129), distribs.Continuous('c0', 190, 256)]


@pytest.mark.parametrize('kind', KINDS)
@pytest.mark.parametrize('positions,reward,success', [
    ([[0.2, 0.2], [0.21, 0.21], [0.8, 0.8], [0.81, 0.81]], 287.5, True),
    ([[0.2, 0.2], [0.25, 0.25], [0.8, 0.8], [0.81, 0.81]], 84.2, True),
    ([[0.2, 0.2], [0.53, 0.53], [0.8, 0.8], [0.81, 0.81]], 0.4, True),
    ([[0.2, 0.53], [0.53, 0.2], [0.8, 0.8], [0.81, 0.81]], 0.4, True),
    ([[0.2, 0.2], [0.53, 0.53], [0.8, 0.8], [0.9, 0.9]], -1.2, False),
    ([[0.2, 0.2], [0.53, 0.53], [0.8, 0.9], [0.9, 0.8]], -1.2, False)])
def test_clustering_4_sprites(kind, positions, reward, success):
  """tests/tasks_test.py:186-208 ClusteringTest.test4Sprites."""
  sprites = [Sprite(x=p[0], y=p[1], c0=c) for p, c in zip(positions, [64, 128, 192, 255])]
  r, s = _task_eval(kind, sprites, tasks.Clustering(cluster_distribs=CLUSTERS))
  assert abs(r - reward) < 0.1 and s == success


@pytest.mark.parametrize('kind', KINDS)
def test_clustering_more_sprites_and_three_clusters(kind):
  """tests/tasks_test.py:210-243 testMoreSprites / test3Clusters."""
  cl = [distribs.Continuous('c0', 50, 100), distribs.Continuous('c0', 200, 250)]
  for p0, p1, want in [
      ([[0.2, 0.2], [0.3, 0.3]], [[0.8, 0.8], [0.8, 0.9], [0.9, 0.9]], 18.7),
      ([[0.2, 0.2], [0.3, 0.3]], [[0.8, 0.8], [0.8, 0.9], [0.9, 0.2]], -2.9),
      ([[0.2, 0.2], [0.3, 0.3], [0.25, 0.3]], [[0.8, 0.8], [0.8, 0.9], [0.9, 0.9]], 21.2),
      ([[0.2, 0.2], [0.3, 0.3], [0.4, 0.8]], [[0.8, 0.8], [0.8, 0.9], [0.9, 0.9]], -1.8)]:
    sprites = [Sprite(x=p[0], y=p[1], c0=75) for p in p0] + [Sprite(x=p[0], y=p[1], c0=225) for p in p1]
    r, _ = _task_eval(kind, sprites, tasks.Clustering(cluster_distribs=cl))
    assert abs(r - want) < 0.1
  sprites = [Sprite(x=0.2, y=0.2, c0=64), Sprite(x=0.3, y=0.3, c0=64), Sprite(x=0.8, y=0.9, c0=128),
             Sprite(x=0.9, y=0.8, c0=128), Sprite(x=0.8, y=0.9, c0=255), Sprite(x=0.9, y=0.8, c0=255)]
  cl3 = [distribs.Continuous('c0', 0, 100), distribs.Continuous('c0', 100, 150), distribs.Continuous('c0', 200, 256)]
  r, _ = _task_eval(kind, sprites, tasks.Clustering(cluster_distribs=cl3))
  assert abs(r - 17.5) < 0.1


@pytest.mark.parametrize('kind', KINDS)
@pytest.mark.parametrize('kwargs,reward', [
    (dict(termination_threshold=2.5), 17.5), (dict(termination_threshold=5.), 5.), (dict(termination_threshold=10.), -20.),
    (dict(termination_threshold=2.5, terminate_bonus=5.), 22.5), (dict(termination_threshold=5., terminate_bonus=3.), 8.),
    (dict(termination_threshold=10., terminate_bonus=7.), -20.),
    (dict(termination_threshold=2.5, reward_range=5.), 8.8), (dict(termination_threshold=5., reward_range=3.), 1.5),
    (dict(termination_threshold=10., reward_range=7.), -14.),
    (dict(termination_threshold=2.5, sparse_reward=True), 17.5), (dict(termination_threshold=7., sparse_reward=True), 0.),
    (dict(termination_threshold=5., sparse_reward=True, terminate_bonus=5.), 10.)])
def test_clustering_parameter_tables(kind, kwargs, reward):
  """tests/tasks_test.py:245-294 testTerminationThreshold / Bonus / RewardRange / SparseReward."""
  sprites = [Sprite(x=0.2, y=0.2, c0=64), Sprite(x=0.3, y=0.3, c0=128), Sprite(x=0.8, y=0.9, c0=192),
             Sprite(x=0.9, y=0.8, c0=255)]
  r, _ = _task_eval(kind, sprites, tasks.Clustering(cluster_distribs=CLUSTERS, **kwargs))
  assert abs(r - reward) < 0.1


def _meta_fixture():
  subtasks = [
      tasks.FindGoalPosition(filter_distrib=distribs.Continuous('c0', 0, 100), goal_position=np.array([0.2, 0.2]),
                             terminate_distance=0.1),
      tasks.FindGoalPosition(filter_distrib=distribs.Continuous('c0', 100, 200), goal_position=np.array([0.5, 0.5]),
                             terminate_distance=0.1, terminate_bonus=5.0),
      tasks.FindGoalPosition(filter_distrib=distribs.Continuous('c0', 200, 256), goal_position=np.array([0.8, 0.8]),
                             terminate_distance=0.1, terminate_bonus=10.0)]
  ok = [Sprite(x=0.2, y=0.2, c0=50), Sprite(x=0.5, y=0.45, c0=150), Sprite(x=0.85, y=0.75, c0=250)]
  bad = [Sprite(x=0.2, y=0.8, c0=50), Sprite(x=0.3, y=0.45, c0=150), Sprite(x=0.9, y=0.75, c0=250)]
  return subtasks, ok, [5., 7.5, 11.5], bad, [-25., -5.3, -0.6]


def _meta_case(successes):
  subtasks, ok, ok_r, bad, bad_r = _meta_fixture()
  si = [i for i, s in enumerate(successes) if s]
  fi = [i for i, s in enumerate(successes) if not s]
  return subtasks, [ok[i] for i in si] + [bad[i] for i in fi], [ok_r[i] for i in si] + [bad_r[i] for i in fi]


@pytest.mark.parametrize('kind', KINDS)
@pytest.mark.parametrize('successes', [(True, True, True), (True, True, False), (True, False, False), (False, False, False)])
def test_meta_aggregated(kind, successes):
  """tests/tasks_test.py:339-392 MetaAggregatedTest.testSum / testMax / testMin / testMean."""
  subtasks, sprites, rewards = _meta_case(successes)
  for agg, want in [('sum', sum(rewards)), ('max', max(rewards)), ('min', min(rewards)), ('mean', np.mean(rewards))]:
    for crit in ('all', 'any'):
      r, s = _task_eval(kind, sprites, tasks.MetaAggregated(subtasks, reward_aggregator=agg, termination_criterion=crit))
      assert abs(r - want) < 0.1, (agg, r, want)
      assert s == (all(successes) if crit == 'all' else any(successes))


@pytest.mark.parametrize('kind', KINDS)
@pytest.mark.parametrize('successes,agg,bonus,reward', [
    ((True, True, True), 'sum', 0., 24.), ((True, True, True), 'sum', 5., 29.), ((True, True, False), 'sum', 5., 11.9),
    ((True, True, True), 'min', 0., 5.), ((True, True, True), 'min', 5., 10.), ((True, True, False), 'min', 5., -0.6)])
def test_meta_aggregated_terminate_bonus(kind, successes, agg, bonus, reward):
  """tests/tasks_test.py:394-409 testTerminateBonus."""
  subtasks, sprites, _ = _meta_case(successes)
  r, _ = _task_eval(kind, sprites, tasks.MetaAggregated(subtasks, reward_aggregator=agg, terminate_bonus=bonus))
  assert abs(r - reward) < 0.1


def test_meta_aggregated_rejects_unknown_names():
  with pytest.raises(ValueError):
    tasks.MetaAggregated([], reward_aggregator='median')
  with pytest.raises(ValueError):
    tasks.MetaAggregated([], termination_criterion='most')


# --------------------------------------------------------------------------- sprite geometry
@pytest.mark.parametrize('x,y,shape,angle,scale,containment', [
    (0.5, 0.5, 'square', 0, 0.5, ['0000', '0110', '0110', '0000']),
    (0.5, 0.5, 'square', 45, 1, ['0110', '1111', '1111', '0110']),
    (0.75, 0.75, 'square', 0, 0.5, ['0011', '0011', '0000', '0000']),
    (0.65, 0.55, 'triangle', 0, 0.5, ['0010', '0010', '0111', '0000']),
    (0.37, 0.55, 'star_5', 0, 0.6, ['0100', '1110', '0100', '0000'])])
def test_contains_point_grids(x, y, shape, angle, scale, containment):
  """tests/sprite_test.py:53-125 SpriteTest.testContainsPoint (hit-test used by the action spaces)."""
  from oracle import oracle
  from spriteworld_amd import shapes
  want = np.flipud(np.array([[c == '1' for c in row] for row in containment]))
  lin = np.linspace(0.1, 0.9, 4)
  got = np.array([[oracle.contains_point(shapes.shape_index(shape), scale, angle, px - x, py - y) for px in lin]
                  for py in lin])
  assert np.array_equal(got, want)


@pytest.mark.parametrize('kind', KINDS)
def test_click_grid_moves_exactly_the_contained_points(kind):
  """Same grids through the engine: a SelectMove click moves the sprite iff the point is inside."""
  for x, y, shape, angle, scale, containment in [
      (0.5, 0.5, 'square', 45, 1, ['0110', '1111', '1111', '0110']),
      (0.37, 0.55, 'star_5', 0, 0.6, ['0100', '1110', '0100', '0000'])]:
    want = np.flipud(np.array([[c == '1' for c in row] for row in containment]))
    lin = np.linspace(0.1, 0.9, 4)
    h = Harness(kind, [Sprite(x=x, y=y, shape=shape, angle=angle, scale=scale)],
                action_space=action_spaces.SelectMove(scale=0.1))
    h.step(NOOP)
    for iy, py in enumerate(lin):
      for ix, px in enumerate(lin):
        h.set_positions([[x, y]])
        moved = not np.allclose(h.step(np.array([px, py, 1.0, 1.0]))['pos'], [[x, y]])
        assert moved == want[iy, ix], (shape, px, py)


# --------------------------------------------------------------------------- renderer
def _render_sprites():
  return [Sprite(x=0.75, y=0.95, shape='spoke_6', scale=0.2, c0=20, c1=50, c2=80),
          Sprite(x=0.2, y=0.3, shape='triangle', scale=0.1, c0=150, c1=255, c2=100),
          Sprite(x=0.7, y=0.5, shape='square', scale=0.3, c0=0, c1=255, c2=0),
          Sprite(x=0.5, y=0.5, shape='square', scale=0.3, c0=255, c1=0, c2=0)]


def _render(kind, sprites, **kw):
  h = Harness(kind, sprites, rends={'image': renderers.PILRenderer(**kw)})
  return h.step(NOOP)['obs']


@pytest.mark.parametrize('kind', KINDS)
def test_pil_renderer_pixels(kind):
  """tests/renderers/pil_renderer_test.py:49-88 background, occlusion, anti-aliasing, colour map."""
  image = _render(kind, _render_sprites(), image_size=(64, 64), bg_color=(5, 6, 7))
  assert list(image[5, 5]) == [5, 6, 7]
  image = _render(kind, _render_sprites(), image_size=(64, 64))
  assert list(image[32, 32]) == [255, 0, 0] and list(image[32, 50]) == [0, 255, 0]
  image = _render(kind, _render_sprites(), image_size=(16, 16), anti_aliasing=5)
  assert list(image[4, 6]) == [0, 0, 0] and list(image[6, 6]) == [255, 0, 0]
  assert all(image[5, 6] >= [50, 0, 0]) and all(image[5, 6] <= [120, 30, 0])
  assert all(image[7, 6] >= [200, 0, 0]) and all(image[7, 6] <= [255, 50, 0])
  image = _render(kind, _render_sprites(), image_size=(16, 16), anti_aliasing=1)
  assert list(image[4, 6]) == [0, 0, 0] and list(image[6, 6]) == [255, 0, 0] and list(image[7, 6]) == [255, 0, 0]
  s = Sprite(x=0.5, y=0.5, shape='square', c0=0.2, c1=0.5, c2=0.5)
  image = _render(kind, [s], image_size=(64, 64), color_to_rgb=renderers.hsv_to_rgb)
  assert list(image[32, 32]) == [114, 127, 63]


# --------------------------------------------------------------------------- environment
@pytest.mark.parametrize('kind', KINDS)
def test_max_episode_length_state_machine(kind):
  """tests/environment_test.py:53-65 testMaxEpisodeLength: FIRST, MIDx6, LAST, FIRST, ..."""
  h = Harness(kind, [Sprite(c0=255)], max_episode_length=7, keep_in_frame=True)
  action = np.array([0.5, 0.5, 0.5, 0.5])
  assert h.step(action)['step_type'] == 0
  for _ in range(3):
    for _ in range(6):
      assert h.step(action)['step_type'] == 1
    assert h.step(action)['step_type'] == 2
    assert h.step(action)['step_type'] == 0


@pytest.mark.parametrize('kind', KINDS)
def test_task_termination_and_auto_reset(kind):
  """tests/environment_test.py:67-88 testTaskTermination."""
  h = Harness(kind, [Sprite(x=0.25, y=0.25, c0=255)], task=tasks.FindGoalPosition(goal_position=(0.5, 0.5)),
              keep_in_frame=True)
  donothing, success = np.array([0.25, 0.25, 0.5, 0.5]), np.array([0.25, 0.25, 0.75, 0.75])
  assert h.step(donothing)['step_type'] == 0
  assert h.step(donothing)['step_type'] == 1
  assert h.step(success)['step_type'] == 2
  assert h.step(success)['step_type'] == 0


# --------------------------------------------------------------------------- handcrafted renderers
def test_sprite_factors_rejects_unknown_factors():
  # tests/renderers/handcrafted_test.py:35-40
  renderers.SpriteFactors(factors=('x', 'y', 'scale'))
  with pytest.raises(ValueError):
    renderers.SpriteFactors(factors=('position', 'scale'))
  with pytest.raises(ValueError):
    renderers.SpriteFactors(factors=('x', 'y', 'size'))


def _factor_rows(sprites):
  from spriteworld_amd import engine
  rends = {'factors': renderers.SpriteFactors()}
  h = Harness('hip', sprites, rends=rends)
  h.step(NOOP)
  return h.eng.factors().cpu().numpy()[0]


@pytest.mark.gpu
@pytest.mark.parametrize('x,y,shape,c0,c1,c2,scale,angle', [
    (0.5, 0.5, 'square', 0, 0, 255, 0.5, 0),
    (0.5, 0.5, 'square', 255, 0, 0, 0.5, 0),
    (0.5, 0.8, 'octagon', 0.4, 0.8, 0.5, 0.6, 90),
    (0.5, 0.3, 'star_5', 180, 180, 0, 0.2, 240),
])
def test_sprite_factors_singleton(x, y, shape, c0, c1, c2, scale, angle):
  # tests/renderers/handcrafted_test.py:92-108
  from spriteworld_amd import shapes as shape_lib
  from spriteworld_amd.sprite import FACTOR_NAMES
  row = _factor_rows([Sprite(x=x, y=y, shape=shape, c0=c0, c1=c1, c2=c2, scale=scale, angle=angle)])[0]
  out = dict(zip(FACTOR_NAMES, row))
  assert out['shape'] == shape_lib.ShapeType[shape].value
  for name, value in (('x', x), ('y', y), ('c0', c0), ('c1', c1), ('c2', c2), ('scale', scale), ('angle', angle)):
    assert abs(out[name] - value) <= 1e-4, name


@pytest.mark.gpu
def test_sprite_factors_two_sprites():
  # tests/renderers/handcrafted_test.py:110-144
  from spriteworld_amd import shapes as shape_lib
  from spriteworld_amd.sprite import FACTOR_NAMES
  vals = dict(x=[0.5, 0.3], y=[0.4, 0.8], shape=['square', 'spoke_4'], c0=[0, 200], c1=[255, 100], c2=[0, 200],
              scale=[0.2, 0.3], angle=[0, 120], x_vel=[0.0, 0.1], y_vel=[-0.2, 0.05])
  rows = _factor_rows([Sprite(**{k: v[i] for k, v in vals.items()}) for i in range(2)])
  for i in range(2):
    out = dict(zip(FACTOR_NAMES, rows[i]))
    assert out['shape'] == shape_lib.ShapeType[vals['shape'][i]].value
    for name in ('x', 'y', 'c0', 'c1', 'c2', 'scale', 'angle', 'x_vel', 'y_vel'):
      assert abs(out[name] - vals[name][i]) <= 1e-4, name

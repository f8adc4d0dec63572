"""Host-side mirrors, lowering, and the dm_env-style environment wrappers."""
import numpy as np
import pytest

from spriteworld_amd import _abi, action_spaces, lanczos, lowering, renderers, sprite_generators, tasks
from spriteworld_amd import factor_distributions as distribs
from spriteworld_amd.sprite import Sprite


def _cobra_like_config(n_targets=2, n_distractors=1):
  shared = distribs.Product([
      distribs.Continuous('x', 0.1, 0.9), distribs.Continuous('y', 0.1, 0.9),
      distribs.Discrete('shape', ['square', 'triangle', 'circle']), distribs.Discrete('scale', [0.13]),
      distribs.Continuous('c1', 0.3, 1.), distribs.Continuous('c2', 0.9, 1.)])
  target_hue, distractor_hue = distribs.Continuous('c0', 0., 0.4), distribs.Continuous('c0', 0.5, 0.9)
  gen = sprite_generators.shuffle(sprite_generators.chain_generators(
      sprite_generators.generate_sprites(distribs.Product([target_hue, shared]), num_sprites=n_targets),
      sprite_generators.generate_sprites(distribs.Product([distractor_hue, shared]), num_sprites=n_distractors)))
  return {
      'task': tasks.FindGoalPosition(filter_distrib=target_hue, terminate_distance=0.075),
      'action_space': action_spaces.SelectMove(scale=0.25),
      'renderers': {'image': renderers.PILRenderer(image_size=(64, 64), anti_aliasing=5,
                                                   color_to_rgb=renderers.hsv_to_rgb)},
      'init_sprites': gen,
      'max_episode_length': 20,
      'metadata': {'name': 'test', 'mode': 'train'},
  }


def test_sprite_factors_and_position_dtype():
  s = Sprite(x=np.float32(0.25), y=np.float32(0.5), shape='star_5', angle=30, scale=0.2, c0=0.1, c1=0.2, c2=0.3)
  assert s.position.dtype == np.float32 and list(s.factors)[:3] == ['x', 'y', 'shape']
  assert Sprite(x=0.3, y=0.4).position.dtype == np.float64
  with pytest.raises(KeyError):
    Sprite(shape='blob')


def test_distributions_sample_and_contain():
  np.random.seed(0)
  c = distribs.Continuous('c0', 0.2, 0.4)
  v = c.sample()
  assert v['c0'].dtype == np.float32 and c.contains(v) and not c.contains({'c0': 0.4})
  p = distribs.Product([c, distribs.Discrete('shape', ['square'])])
  assert p.contains({'c0': 0.3, 'shape': 'square'}) and not p.contains({'c0': 0.3, 'shape': 'circle'})
  m = distribs.Mixture([distribs.Continuous('c0', 0, 0.1), distribs.Continuous('c0', 0.9, 1.0)])
  assert m.contains({'c0': 0.95}) and not m.contains({'c0': 0.5})
  sm = distribs.SetMinus(distribs.Continuous('c0', 0, 1), distribs.Continuous('c0', 0.2, 0.8))
  for _ in range(20):
    x = sm.sample()['c0']
    assert not 0.2 <= x < 0.8
  with pytest.raises(ValueError):
    distribs.Product([c, distribs.Continuous('c0', 0, 1)])


def test_lower_config_and_episodes():
  np.random.seed(1)
  config = _cobra_like_config()
  episodes = [config['init_sprites']() for _ in range(6)]
  assert lowering.position_dtype(episodes) == np.float32
  cfg = lowering.lower_config(config['task'], config['action_space'], config['renderers'], True, 20, 2, 3, True)
  assert (cfg.image_h, cfg.image_w, cfg.anti_aliasing, cfg.n_tasks, cfg.is_meta) == (64, 64, 5, 1, 0)
  assert cfg.action_space == _abi.ACTION_SELECT_MOVE and cfg.action_scale == 0.25
  assert cfg.tasks[0].kind == _abi.TASK_FIND_GOAL and cfg.tasks[0].terminate_distance == 0.075
  pool = lowering.lower_episodes(episodes, config['task'], config['renderers'], max_sprites=3)
  pool.assign_round_robin(2, 3)
  assert pool.x.shape == (6, 3) and list(pool.n_sprites) == [3] * 6
  for e, ep in enumerate(episodes):
    for s, sp in enumerate(ep):
      assert pool.label[e, 0, s] == int(sp.c0 < 0.4)
      assert tuple(pool.rgb[e, s, :3]) == renderers.hsv_to_rgb(sp.color)
      assert pool.x[e, s] == float(sp.position[0])
  assert list(pool.pool_base) == [0, 3] and list(pool.pool_len) == [3, 3]
  pool.as_struct()


def test_lowering_rejects_what_the_engine_cannot_do():
  config = _cobra_like_config()
  with pytest.raises(lowering.LoweringError):
    lowering.lower_config(config['task'], config['action_space'], config['renderers'], max_sprites=99)
  with pytest.raises(lowering.LoweringError):
    lowering.position_dtype([[Sprite(x=0.1, y=0.2), Sprite(x=np.float32(0.1), y=np.float32(0.2))]])


def test_meta_task_lowering():
  subs = [tasks.FindGoalPosition(goal_position=(0.75, 0.25)), tasks.Clustering([None, None])]
  cfg = lowering.lower_config(tasks.MetaAggregated(subs, 'mean', 'any', terminate_bonus=3.),
                              action_spaces.Embodied(step_size=0.1, motion_cost=2.), {}, max_sprites=4)
  assert (cfg.is_meta, cfg.n_tasks, cfg.meta_aggregator, cfg.meta_termination) == (1, 2, _abi.AGG_MEAN, _abi.TERM_ANY)
  assert cfg.meta_terminate_bonus == 3. and cfg.action_space == _abi.ACTION_EMBODIED and cfg.action_scale == 0.1
  assert cfg.tasks[0].goal_position[0] == 0.75 and cfg.tasks[1].kind == _abi.TASK_CLUSTERING


def test_lanczos_tables_properties():
  bounds, coeffs = lanczos.resample_tables(320, 64)
  assert coeffs.shape == (64, 31)
  assert np.all(bounds[3:61, 1] == 30) and np.all(bounds[3:61, 0] == 5 * np.arange(3, 61) - 12)
  assert len({tuple(r) for r in coeffs[3:61]}) == 1          # shift-invariant interior
  assert abs(int(coeffs[10].sum()) - (1 << 22)) <= 2


@pytest.mark.gpu
def test_batched_environment_runs_a_config():
  import torch
  from spriteworld_amd import environment
  np.random.seed(2)
  config = _cobra_like_config()
  config['renderers']['success'] = renderers.Success()
  env = environment.BatchedEnvironment(num_envs=64, episodes_per_env=4, **config)
  ts = env.reset()
  assert ts.step_type.shape == (64,) and int(ts.step_type.max()) == 0
  assert ts.observation['image'].shape == (64, 64, 64, 3) and ts.observation['image'].dtype == torch.uint8
  assert set(env.observation_spec()) == {'image', 'success'}
  firsts = lasts = 0
  for _ in range(60):
    ts = env.step(env.sample_actions())
    firsts += int((ts.step_type == 0).sum())
    lasts += int((ts.step_type == 2).sum())
    mid = ts.step_type != 0
    assert not torch.isnan(ts.reward[mid]).any() and torch.isnan(ts.reward[~mid]).all()
  assert lasts >= 64 * 2 and abs(firsts - lasts) <= 64     # 20-step episodes auto-reset
  env.refill_pool()
  assert int(env.step(env.sample_actions()).step_type.max()) == 0
  env.close()


@pytest.mark.gpu
def test_single_environment_follows_example_run_loop():
  """example_run_loop.py:62-80: reset(), step(action_space.sample()) until last(), log success."""
  from spriteworld_amd import environment
  np.random.seed(3)
  config = _cobra_like_config()
  config['renderers']['success'] = renderers.Success()
  env = environment.Environment(**config)
  for _ in range(3):
    timestep = env.reset()
    assert timestep.first() and timestep.reward is None and timestep.discount is None
    rewards, n = [], 0
    while not timestep.last():
      timestep = env.step(env.action_space.sample())
      rewards.append(timestep.reward)
      n += 1
    assert n <= 20 and timestep.discount == 0.0
    assert isinstance(timestep.observation['success'], bool)
    assert timestep.observation['image'].shape == (64, 64, 3) and np.isfinite(np.nanmean(rewards))
  env.close()


@pytest.mark.gpu
def test_sprite_factors_observation_and_action_noise():
  """handcrafted.SpriteFactors as a batched tensor, and SelectMove(noise_scale=...) noise."""
  import torch
  from spriteworld_amd import environment, shapes
  np.random.seed(4)
  config = _cobra_like_config()
  config['renderers'] = {'factors': renderers.SpriteFactors(), 'xy': renderers.SpriteFactors(factors=('y', 'x', 'shape'))}
  config['action_space'] = action_spaces.SelectMove(scale=0.25, noise_scale=0.05)
  env = environment.BatchedEnvironment(num_envs=32, episodes_per_env=2, device_reset=False, **config)   # host pool: compared below
  env.seed_noise(0)
  ts = env.reset()
  f = ts.observation['factors'].cpu().numpy()
  assert f.shape == (32, 3, 10) and ts.observation['xy'].shape == (32, 3, 3)
  pool, st = env.engine.pool, env.state()
  for n in range(32):
    e = st['pool_entry'][n]
    assert np.array_equal(f[n, :, 0], st['x'][n]) and np.array_equal(f[n, :, 1], st['y'][n])
    assert np.array_equal(f[n, :, 2], pool.shape[e] + 1) and np.array_equal(f[n, :, 4], pool.scale[e])
    assert np.array_equal(f[n, :, 5:8], pool.color[e]) and np.array_equal(f[n, :, 3], pool.angle[e])
  assert np.array_equal(ts.observation['xy'].cpu().numpy(), f[:, :, [1, 0, 2]])
  # noise: the same clean action moves sprites by different amounts in different environments
  a = np.tile(np.array([[0.5, 0.5, 0.9, 0.9]]), (32, 1))
  env.engine.set_positions(np.full((32, 3), 0.5), np.full((32, 3), 0.5))
  ts = env.step(a)
  moved = ts.observation['factors'][:, 2, 0].cpu().numpy() - 0.5
  hit = moved != 0                      # a noised click may miss the sprite
  assert hit.sum() >= 8 and np.all(np.abs(moved[hit] - 0.1) < 0.1) and np.std(moved[hit]) > 1e-3
  env.close()


@pytest.mark.gpu
@pytest.mark.parametrize('action_dtype', [np.float64, np.float32])
def test_select_move_noise_is_gaussian_with_the_given_scale(action_dtype):
  """action_spaces.py:69-75: `action + np.random.normal(loc=0, scale=noise_scale, size=4)`.  One big square in the middle,
  a click on it and a fixed motion: the displacement of the sprite, divided by the motion scale, is the noise of the
  motion component -- Kolmogorov-Smirnov against N(0, noise_scale) on 2048 environments, for float64 and float32
  actions (numpy promotes the noised float32 action to float64: the engine then runs its float64-action arithmetic)."""
  import scipy.stats
  from spriteworld_amd import environment
  sigma, scale = 0.05, 0.25
  env = environment.BatchedEnvironment(
      task=tasks.NoReward(), action_space=action_spaces.SelectMove(scale=scale, noise_scale=sigma),
      renderers={'image': renderers.PILRenderer(image_size=(64, 64), anti_aliasing=1)},
      init_sprites=lambda: [Sprite(x=0.5, y=0.5, shape='square', scale=0.6, c0=255, c1=0, c2=0)],
      max_episode_length=1000, num_envs=2048, episodes_per_env=1, device_reset=False, action_dtype=action_dtype)
  assert env._cfg.action_is_f32 == 0            # pylint: disable=protected-access
  env.seed_noise(123)
  env.reset()
  a = np.tile(np.array([[0.5, 0.5, 0.9, 0.3]], dtype=action_dtype), (2048, 1))
  ts = env.step(a)
  st = env.state()
  dx, dy = st['x'][:, 0] - 0.5, st['y'][:, 0] - 0.5
  assert np.all(dx != 0)                         # the click (0.5 +- 0.05) never misses a square of side 0.6
  for d, clean in ((dx, 0.9), (dy, 0.3)):
    noise = d / scale - (np.float64(action_dtype(clean)) - 0.5)
    assert abs(noise.mean()) < 4 * sigma / np.sqrt(2048) and abs(noise.std() / sigma - 1) < 0.08
    assert scipy.stats.kstest(noise, 'norm', args=(0, sigma)).pvalue > 1e-3
  assert ts.reward.dtype == __import__('torch').float64
  env.close()


@pytest.mark.gpu
def test_environment_groups_equal_the_groups_stepped_alone():
  """EnvironmentGroups: per-group streams change the schedule, not the results."""
  import torch
  from spriteworld_amd import environment
  config = _cobra_like_config()
  np.random.seed(11)
  groups = environment.EnvironmentGroups(num_groups=2, num_envs=64, device_reset=False, **config)
  np.random.seed(11)
  alone = [environment.BatchedEnvironment(num_envs=32, device_reset=False, **config) for _ in range(2)]
  gen = torch.Generator().manual_seed(4)
  for g in range(2):
    a, b = groups.reset(g), alone[g].reset()
    groups.synchronize()
    assert torch.equal(a.observation['image'], b.observation['image'])
  for _ in range(12):
    acts = [torch.rand((32, 4), generator=gen, dtype=torch.float64).cuda() for _ in range(2)]
    torch.cuda.synchronize()
    outs = [groups.step(g, acts[g]) for g in range(2)]
    groups.synchronize()
    for g in range(2):
      ref = alone[g].step(acts[g])
      torch.cuda.synchronize()
      assert torch.equal(outs[g].observation['image'], ref.observation['image'])
      np.testing.assert_array_equal(outs[g].reward.cpu().numpy(), ref.reward.cpu().numpy())
      assert torch.equal(outs[g].step_type, ref.step_type)
  groups.close()
  for e in alone:
    e.close()

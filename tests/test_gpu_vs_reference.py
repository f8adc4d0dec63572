"""The UNMODIFIED reference and the HIP engine in ONE process on the GPU node (the direct leg of the parity chain).

Everywhere else parity is transitive (reference -> oracle / golden in the build container; oracle / golden -> HIP on the
GPU box).  Here the reference's own `Environment` (`/root/reference/spriteworld/environment.py:88-108`; on the GPU box the
sourceless bytecode of exactly those modules under `oracle/_ref`, see oracle/stage_ref.py) is stepped beside
`spriteworld_amd.engine.Engine` -- the C ABI, the HIP kernels -- on the same lowered episodes and the same actions, for
every shipped config in both modes: step types, rewards, success and sprite positions bit-exact, frames +-0 (north_star
allows +-1 LSB).  The body is `tests/test_oracle_vs_reference.py::test_oracle_equals_reference_environment` with the
engine in the oracle's place.
"""
import copy
import importlib

import numpy as np
import pytest

from oracle import ref_harness

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not ref_harness.reference_available(),
                                 reason='neither /root/reference nor oracle/_ref (python oracle/stage_ref.py) present')]

_MODULES = [
    'spriteworld.configs.cobra.goal_finding_new_position',
    'spriteworld.configs.cobra.goal_finding_new_shape',
    'spriteworld.configs.cobra.goal_finding_more_distractors',
    'spriteworld.configs.cobra.goal_finding_more_targets',
    'spriteworld.configs.cobra.clustering',
    'spriteworld.configs.cobra.sorting',
    'spriteworld.configs.cobra.exploration',
    'spriteworld.configs.examples.goal_finding_embodied',
    'spriteworld.configs.examples.goal_finding_clustering',
]
CONFIGS = [(m, mode) for m in _MODULES for mode in ('train', 'test')]


def _fresh_episodes(episodes):
  """The reference's init_sprites following the pool: the constructor's own draw (environment.py:68), then the episodes in
  order, wrapping around, as NEW sprite objects every time."""
  yield copy.deepcopy(episodes[0])
  while True:
    for e in episodes:
      yield copy.deepcopy(e)


def _bits(v):
  return np.float64(v).view(np.uint64)


def test_third_party_libraries_of_the_reference_import_here():
  """The arithmetic the reference delegates must be the versions the oracle was pinned against."""
  v = ref_harness.third_party_versions()
  assert not any(str(x).startswith('missing') for x in v.values()), v
  print('reference runs as', ref_harness.reference_kind(), 'from', ref_harness.REFERENCE_ROOT, v)


@pytest.mark.parametrize('module,mode', CONFIGS)
def test_hip_engine_equals_reference_environment(module, mode):
  ref_harness.load_reference()
  from spriteworld import environment
  from spriteworld import renderers as ref_renderers
  from spriteworld_amd import engine, lowering
  seed, n_eps, n_steps = 21, 30, 250
  np.random.seed(seed)
  config = importlib.import_module(module).get_config(mode)
  episodes = [config['init_sprites']() for _ in range(n_eps)]
  task, aspace, rends = config['task'], config['action_space'], config['renderers']
  S = max(len(e) for e in episodes)
  cfg = lowering.lower_config(task, aspace, rends, True, config['max_episode_length'], 1, S,
                              pos_is_f32=(lowering.position_dtype(episodes) == np.float32))
  pool = lowering.lower_episodes(episodes, task, rends, max_sprites=S).assign_round_robin(1)
  eng = engine.Engine(cfg, pool)
  it = _fresh_episodes(episodes)
  config = dict(config, init_sprites=lambda: next(it))
  config['renderers'] = dict(rends, success=ref_renderers.Success())
  env = environment.Environment(**config)
  rng = np.random.RandomState(seed + 1)
  for t in range(n_steps):
    if cfg.action_space == 2:
      a = np.array([rng.randint(0, 2), rng.randint(0, 4)])
      ts = env.step([int(a[0]), int(a[1])])
    else:
      a = rng.uniform(0, 1, 4)
      ts = env.step(a)
    eng.step(a[None])
    out = eng.outputs_host()
    assert not out['error'][0], t
    assert int(ts.step_type) == int(out['step_type'][0]), t
    r = np.nan if ts.reward is None else float(ts.reward)
    assert (np.isnan(r) and np.isnan(out['reward'][0])) or _bits(r) == _bits(out['reward'][0]), (t, r, out['reward'][0])
    assert bool(ts.observation['success']) == bool(out['success'][0]), t
    assert np.array_equal(ts.observation['image'], out['obs'][0]), t
    st = eng.state()
    pos = np.array([s.position for s in env._sprites], dtype=np.float64).reshape(-1, 2)
    n = st['n_sprites'][0]
    assert n == len(pos)
    assert np.array_equal(pos[:, 0], st['x'][0, :n]) and np.array_equal(pos[:, 1], st['y'][0, :n]), t
  eng.close()


def test_hip_engine_equals_reference_on_a_batch_of_reference_environments():
  """N = 48 reference environments (the headline scene's generators, np.random.seed(1000 + i) as SURVEY 8d prescribes)
  beside ONE batched engine launch per step."""
  ref_harness.load_reference()
  from spriteworld import environment, sprite_generators, tasks
  from spriteworld import factor_distributions as distribs
  from spriteworld import renderers as ref_renderers
  from spriteworld.configs.cobra import common
  from spriteworld_amd import engine, lowering
  N, EPS, STEPS = 48, 6, 60
  clusters = [distribs.Continuous('c0', 0.55, 0.65), distribs.Continuous('c0', 0.27, 0.37)]
  other = distribs.Product([
      distribs.Continuous('x', 0.1, 0.9), distribs.Continuous('y', 0.1, 0.9),
      distribs.Discrete('shape', ['square', 'triangle', 'circle']), distribs.Discrete('scale', [0.13]),
      distribs.Continuous('c1', 0.3, 1.), distribs.Continuous('c2', 0.9, 1.)])
  gens = [sprite_generators.generate_sprites(distribs.Product((other, c0)), num_sprites=n) for c0, n in zip(clusters, (2, 3))]
  gen = sprite_generators.shuffle(sprite_generators.chain_generators(*gens))
  task = tasks.Clustering(clusters, terminate_bonus=0., reward_range=10.)
  aspace, rends = common.action_space(), common.renderers()
  per_env, envs = [], []
  for i in range(N):
    np.random.seed(1000 + i)
    eps = [gen() for _ in range(EPS)]
    per_env.append(eps)
    it = _fresh_episodes(eps)
    envs.append(environment.Environment(task=task, action_space=aspace, init_sprites=(lambda it=it: next(it)),
                                        renderers=dict(rends, success=ref_renderers.Success()), max_episode_length=12))
  flat = [e for eps in per_env for e in eps]
  cfg = lowering.lower_config(task, aspace, rends, True, 12, N, 5, pos_is_f32=True)
  pool = lowering.lower_episodes(flat, task, rends, max_sprites=5)
  pool.pool_base = np.arange(N, dtype=np.int32) * EPS
  pool.pool_len = np.full(N, EPS, np.int32)
  eng = engine.Engine(cfg, pool)
  from spriteworld_amd import _abi
  dropped = set()        # environments in which the reference raised: compared no further (the others go on to the last step)
  for t in range(STEPS):
    acts = np.random.RandomState(2000 + t).uniform(size=(N, 4))
    eng.step(acts)
    out = eng.outputs_host()
    st = eng.state()
    for i, env in enumerate(envs):
      if i in dropped:
        continue
      try:
        ts = env.step(acts[i])
      except ZeroDivisionError:      # tasks.py:215 1. / 0. (collapsed clusters): the engine flags exactly that environment
        assert out['error'][i] & _abi.ENV_ERR_DB_ZERO, (t, i)
        dropped.add(i)
        continue
      assert not out['error'][i], (t, i)
      assert int(ts.step_type) == int(out['step_type'][i]), (t, i)
      r = np.nan if ts.reward is None else float(ts.reward)
      assert (np.isnan(r) and np.isnan(out['reward'][i])) or _bits(r) == _bits(out['reward'][i]), (t, i)
      assert np.array_equal(ts.observation['image'], out['obs'][i]), (t, i)
      pos = np.array([s.position for s in env._sprites], dtype=np.float64).reshape(-1, 2)
      assert np.array_equal(pos[:, 0], st['x'][i, :5]) and np.array_equal(pos[:, 1], st['y'][i, :5]), (t, i)
  assert len(dropped) < N // 4, sorted(dropped)
  eng.close()

"""The N = 1 `Environment`'s remaining reference surface -- `observation()`, `success()`, `should_terminate()`,
`sample_contained_position()`, `state()` (/root/reference/spriteworld/environment.py:80-86,110-142) -- as known-answer tests
against the UNMODIFIED reference class stepped beside it on the same episodes, actions and numpy random stream.

Backends: `oracle` (tests/_fake_engine.py) and `emulated_kernel` (the kernel source on the host) in the build container,
`hip` (`-m gpu`, the product on the MI355X; the reference from oracle/_ref).
"""
import copy

import numpy as np
import pytest

from oracle import ref_harness

pytestmark = pytest.mark.skipif(not ref_harness.reference_available(),
                                reason='neither /root/reference nor oracle/_ref (python oracle/stage_ref.py) present')

BACKENDS = ['oracle', 'emulated_kernel', pytest.param('hip', marks=pytest.mark.gpu)]


def _use_backend(monkeypatch, backend):
  from spriteworld_amd import environment as amd_environment
  if backend == 'oracle':
    from tests import _fake_engine
    monkeypatch.setattr(amd_environment._engine, 'Engine', _fake_engine.FakeEngine)
  elif backend == 'emulated_kernel':
    from tests import _emu_engine
    monkeypatch.setattr(amd_environment._engine, 'Engine', _emu_engine.EmuTorchEngine)


def _pair(episodes, metadata=None, max_episode_length=9, embodied=False):
  """(reference Environment, drop-in Environment) over the same replayed episodes."""
  from spriteworld import action_spaces, environment, renderers, tasks
  from spriteworld import factor_distributions as distribs
  from spriteworld_amd import environment as amd_environment
  task = tasks.FindGoalPosition(filter_distrib=distribs.Continuous('c0', 0., 0.5), goal_position=(0.5, 0.5), terminate_distance=0.12)
  aspace = action_spaces.Embodied(step_size=0.1) if embodied else action_spaces.SelectMove(scale=0.5)
  rends = {'image': renderers.PILRenderer(image_size=(64, 64), anti_aliasing=5, color_to_rgb=renderers.color_maps.hsv_to_rgb),
           'success': renderers.Success()}
  def cycle(first):
    for e in first:
      yield copy.deepcopy(e)
    while True:
      for e in episodes:
        yield copy.deepcopy(e)

  it_ref = cycle([episodes[0]])       # the same calls for both: the first one is the constructor's (environment.py:68),
  it_our = cycle([episodes[0]])       # which the drop-in keeps as pool entry 0 and never steps
  ref = environment.Environment(task, aspace, rends, lambda: next(it_ref), keep_in_frame=False,
                                max_episode_length=max_episode_length, metadata=metadata)
  ours = amd_environment.Environment(task, aspace, rends, lambda: next(it_our), keep_in_frame=False,
                                     max_episode_length=max_episode_length, metadata=metadata, episodes_per_pool=len(episodes))
  return ref, ours


def _episodes(seed, n=5):
  from spriteworld import sprite
  rng = np.random.RandomState(seed)
  return [[sprite.Sprite(x=np.float32(rng.uniform(0.1, 0.9)), y=np.float32(rng.uniform(0.1, 0.9)),
                         shape=str(rng.choice(['square', 'triangle', 'circle', 'star_5', 'spoke_4'])), angle=float(rng.randint(0, 360)),
                         scale=float(rng.choice([0.1, 0.17, 0.3])), c0=np.float32(rng.uniform(0, 1)), c1=np.float32(0.9),
                         c2=np.float32(1.0), x_vel=float(rng.choice([0., 0.04])), y_vel=float(rng.choice([0., -0.05])))
           for _ in range(int(rng.randint(2, 6)))] for _ in range(n)]


@pytest.mark.parametrize('backend', BACKENDS)
def test_observation_success_and_should_terminate_follow_the_reference(monkeypatch, backend):
  ref_harness.load_reference()
  _use_backend(monkeypatch, backend)
  ref, ours = _pair(_episodes(3), max_episode_length=7)
  rng = np.random.RandomState(8)
  terminated = 0
  for t in range(45):
    a = rng.uniform(0, 1, 4)
    tr, to = ref.step(a), ours.step(a)
    assert int(tr.step_type) == int(to.step_type), t
    obs_r, obs_o = ref.observation(), ours.observation()          # environment.py:136-142, rendered again on demand
    assert set(obs_r) == set(obs_o)
    assert np.array_equal(obs_r['image'], obs_o['image']) and np.array_equal(obs_o['image'], to.observation['image']), t
    assert bool(obs_r['success']) == obs_o['success'], t
    assert ref.success() == ours.success(), t                       # :80-81
    assert bool(ref.should_terminate()) == ours.should_terminate(), t   # :83-86 (timeout, out of frame -- velocities carry sprites out --, success)
    terminated += bool(ref.should_terminate())
  assert terminated >= 4
  ours.close()


@pytest.mark.parametrize('backend', BACKENDS)
def test_success_sees_what_changed_since_the_last_step(monkeypatch, backend):
  """environment.py:80-81 evaluates the task on the sprites as they ARE: positions written or a setter called since the last
  step count (the drop-in returned the last launch's flag)."""
  ref_harness.load_reference()
  _use_backend(monkeypatch, backend)
  from spriteworld import action_spaces, environment, renderers, tasks
  from spriteworld import factor_distributions as distribs
  from spriteworld_amd import environment as amd_environment
  # (a) positions: every sprite put on the goal
  ref, ours = _pair(_episodes(7), max_episode_length=50)
  a = np.array([0., 0., 0.5, 0.5])
  for _ in range(3):
    ref.step(a), ours.step(a)
  assert not ref.success() and not ours.success()
  for s in ref._sprites:
    s._position = np.array([0.5, 0.5])
  st = ours.soa_state()
  ours._batched.engine.set_positions(np.full_like(st['x'], 0.5), np.full_like(st['y'], 0.5))
  assert ref.success() and ours.success()
  assert ref.should_terminate() and ours.should_terminate()
  assert ref.state()['global_state']['success'] and ours.state()['global_state']['success']
  ours.close()
  # (b) a setter under a filter keyed on the attribute it sets: a sprite far from the goal leaves the task's filter
  eps = _episodes(9)
  for e in eps:
    for i, s in enumerate(e):
      s._scale = 0.1
      s._position = np.array([0.5, 0.5]) if i else np.array([0.9, 0.1])
      s._velocity = np.zeros(2)
  task = tasks.FindGoalPosition(filter_distrib=distribs.Continuous('scale', 0.05, 0.2), goal_position=(0.5, 0.5), terminate_distance=0.1)
  rends = {'image': renderers.PILRenderer(image_size=(64, 64), anti_aliasing=5, color_to_rgb=renderers.color_maps.hsv_to_rgb),
           'success': renderers.Success()}
  mk = lambda mod, **kw: mod.Environment(task, action_spaces.SelectMove(scale=0.), rends, (lambda it=iter([copy.deepcopy(e) for e in [eps[0]] + eps]): next(it)),
                                         keep_in_frame=False, max_episode_length=50, **kw)
  ref, ours = mk(environment), mk(amd_environment, episodes_per_pool=len(eps) + 1)
  for _ in range(2):
    ref.step(a), ours.step(a)
  assert not ref.success() and not ours.success()
  assert not ref.observation()['success'] and not ours.observation()['success']
  ref.state()['sprites'][0].scale = 0.4          # sprite.py:166-175; out of the filter: the others are on the goal
  ours.state()['sprites'][0].scale = 0.4
  assert ref.success() and ours.success()
  # observation() goes through state() -> success() on the current sprites too (environment.py:128-142; round-5 advice: the
  # drop-in's Success key carried the last step's flag)
  assert ref.observation()['success'] and ours.observation()['success']
  ours.close()


@pytest.mark.parametrize('backend', BACKENDS)
def test_sample_contained_position_draws_what_the_reference_draws(monkeypatch, backend):
  """environment.py:110-126 / sprite.py:117-126 under the same np.random stream: the same sprite, the same tries, the same point."""
  ref_harness.load_reference()
  _use_backend(monkeypatch, backend)
  ref, ours = _pair(_episodes(11), max_episode_length=50)
  rng = np.random.RandomState(2)
  for t in range(14):
    a = rng.uniform(0, 1, 4)
    ref.step(a), ours.step(a)
    np.random.seed(100 + t)
    want = [ref.sample_contained_position() for _ in range(4)]
    tail_ref = np.random.uniform()
    np.random.seed(100 + t)
    got = [ours.sample_contained_position() for _ in range(4)]
    tail_our = np.random.uniform()
    assert tail_ref == tail_our, t                                    # the same number of draws
    for w, g in zip(want, got):
      assert g.shape == (2,) and g.dtype == np.float64
      assert np.array_equal(np.asarray(w, dtype=np.float64), g), (t, w, g)
      assert any(s.contains_point(g) for s in ours.sprites)
  ours.close()


@pytest.mark.parametrize('backend', BACKENDS)
def test_state_is_the_references_dict(monkeypatch, backend):
  """environment.py:128-134: {'sprites', 'global_state': {'success', 'metadata'}}; the sprites carry the reference's factors."""
  ref_harness.load_reference()
  _use_backend(monkeypatch, backend)
  meta = {'name': 'kat', 'level': 3}
  # before the first step: the reference already holds its constructor's sprites (environment.py:68), the drop-in puts its
  # first episode on the device for that (and, like the reference, draws the next one at the first step)
  ref0, early = _pair(_episodes(5), metadata=meta, max_episode_length=6)
  s0, r0 = early.state(), ref0.state()
  assert set(s0) == {'sprites', 'global_state'} and s0['global_state']['metadata'] == meta
  assert [s.shape for s in s0['sprites']] == [s.shape for s in r0['sprites']]
  assert np.array_equal(early.observation()['image'], ref0.observation()['image'])
  assert early.step(np.zeros(4)).first()
  # ... and having looked changes nothing: the episodes that are stepped are the reference's (looking
  # used to consume an episode, so that every later episode was the next draw)
  assert ref0.step(np.zeros(4)).first()
  rng0 = np.random.RandomState(6)
  for t in range(20):                 # max_episode_length = 6: four episodes
    a = rng0.uniform(0, 1, 4)
    tr, to = ref0.step(a), early.step(a)
    assert int(tr.step_type) == int(to.step_type) and np.array_equal(tr.observation['image'], to.observation['image']), t
  early.close()
  ref, ours = _pair(_episodes(5), metadata=meta, max_episode_length=6)
  rng = np.random.RandomState(4)
  for t in range(15):
    a = rng.uniform(0, 1, 4)
    tr, to = ref.step(a), ours.step(a)
    sr, so = ref.state(), ours.state()
    assert set(sr) == set(so) == {'sprites', 'global_state'}
    assert set(sr['global_state']) == set(so['global_state']) == {'success', 'metadata'}
    assert bool(sr['global_state']['success']) == so['global_state']['success'] and so['global_state']['metadata'] is meta
    assert len(sr['sprites']) == len(so['sprites'])
    for a_, b_ in zip(sr['sprites'], so['sprites']):
      fa, fb = a_.factors, b_.factors
      assert list(fa) == list(fb)
      assert fa['shape'] == fb['shape']
      for k in ('x', 'y', 'angle', 'scale', 'x_vel', 'y_vel'):
        assert float(fa[k]) == float(fb[k]), (t, k)
      assert bool(a_.out_of_frame) == b_.out_of_frame
      assert np.array_equal(a_.vertices, b_.vertices)
  no_meta = _pair(_episodes(5), metadata=None)[1]
  assert set(no_meta.state()['global_state']) == {'success'}         # `if self._metadata:` (:132)
  assert 'x' in no_meta.soa_state()                                  # the structure-of-arrays view keeps its own name
  no_meta.close()
  ours.close()

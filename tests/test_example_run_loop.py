"""The reference's own `example_run_loop.py` (BASELINE configs[0]; example_run_loop.py:62-80) driving the drop-in.

`main()` of the UNMODIFIED script runs twice in this process on the reference's shipped config module
(`spriteworld.configs.cobra.goal_finding_new_position`): once as shipped, once with its `environment` symbol
pointing at `spriteworld_amd.environment` -- the one-line change INTEGRATION.md describes.  The reference's
config dict (its task / action-space / renderer / generator objects) goes unchanged into
`spriteworld_amd.environment.Environment`, is lowered and stepped through the engine interface; the per-episode
log lines (success, mean reward) and every time step must be identical.

Three engines behind the interface: (a) tests/_fake_engine.py, the CPU oracle with the engine's method surface, and
(b) tests/_emu_engine.py, the library's own sources -- C-ABI host side and kernels -- compiled for the host and executed
lane by lane (tests/emu), both in the GPU-less build container; (c) `hip` (`-m gpu`): the product itself --
`spriteworld_amd.engine.Engine`, libswb.so on the MI355X -- driven by the unmodified script on the GPU node, where the
reference is present as the sourceless bytecode of `oracle/_ref` (oracle/stage_ref.py): BASELINE configs[0] end to end.
"""
import copy
import importlib
import logging
import sys
import types

import numpy as np
import pytest

from oracle import ref_harness

pytestmark = pytest.mark.skipif(not ref_harness.reference_available(),
                                reason='neither /root/reference nor oracle/_ref (python oracle/stage_ref.py) present')

CONFIG = 'spriteworld.configs.cobra.goal_finding_new_position'
N_EPISODES = 6


class _Log(logging.Handler):

  def __init__(self):
    logging.Handler.__init__(self)
    self.lines = []

  def emit(self, record):
    self.lines.append(record.getMessage())


def _run_main(monkeypatch, run_loop, episodes, use_dropin, steps):
  """One call of example_run_loop.main() with init_sprites replaying `episodes` (no numpy draws, so that the
  RandomAgent's np.random stream is the same in both runs) and every time step recorded."""
  from spriteworld import environment as ref_environment
  from spriteworld_amd import environment as amd_environment
  real = importlib.import_module(CONFIG)
  it = iter(())

  def get_config(mode):
    nonlocal it
    config = real.get_config(mode)
    # the same calls for both: call 0 is the constructor's (environment.py:68; the drop-in: pool entry 0, never stepped),
    # call k >= 1 the k-th episode (the drop-in draws them up front, as one pool)
    it = (copy.deepcopy(e) for e in [episodes[0]] + episodes)
    config['init_sprites'] = lambda: next(it)
    config['max_episode_length'] = 25
    return config

  fake_module = types.ModuleType('replayed_config')
  fake_module.get_config = get_config
  monkeypatch.setitem(sys.modules, 'replayed_config', fake_module)
  run_loop.FLAGS.config = 'replayed_config'
  run_loop.FLAGS.mode = 'train'
  run_loop.FLAGS.num_episodes = N_EPISODES

  env_module = amd_environment if use_dropin else ref_environment
  base = env_module.Environment

  class Recording(base):      # the loop's own Environment, with its time steps recorded

    def __init__(self, **kwargs):
      if use_dropin:
        kwargs['episodes_per_pool'] = len(episodes) + 1
      base.__init__(self, **kwargs)

    def step(self, action):
      ts = base.step(self, action)
      steps.append((int(ts.step_type), None if ts.reward is None else np.float64(ts.reward).view(np.uint64),
                    bool(ts.observation['success']), ts.observation['image'].copy()))
      return ts

  shim = types.SimpleNamespace(Environment=Recording)
  monkeypatch.setattr(run_loop, 'environment', shim)
  handler = _Log()
  logger = logging.getLogger('absl')
  logger.addHandler(handler)
  logger.setLevel(logging.INFO)
  try:
    np.random.seed(11)
    run_loop.main([])
  finally:
    logger.removeHandler(handler)
  return handler.lines


@pytest.mark.parametrize('backend', ['oracle', 'emulated_kernel', pytest.param('hip', marks=pytest.mark.gpu)])
def test_example_run_loop_main_drives_the_dropin(monkeypatch, backend):
  ref_harness.load_reference()
  from spriteworld_amd import environment as amd_environment
  run_loop = importlib.import_module('example_run_loop')
  if backend == 'oracle':
    from tests import _fake_engine
    monkeypatch.setattr(amd_environment._engine, 'Engine', _fake_engine.FakeEngine)
  elif backend == 'emulated_kernel':       # the kernel source itself, run lane by lane on the host (tests/emu)
    from tests import _emu_engine
    monkeypatch.setattr(amd_environment._engine, 'Engine', _emu_engine.EmuTorchEngine)
  else:                                    # the product: engine.Engine -> libswb.so -> the HIP kernels
    from spriteworld_amd import engine as real_engine
    assert amd_environment._engine.Engine is real_engine.Engine
  np.random.seed(5)
  episodes = [importlib.import_module(CONFIG).get_config('train')['init_sprites']() for _ in range(N_EPISODES + 2)]
  ref_steps, our_steps = [], []
  ref_lines = _run_main(monkeypatch, run_loop, episodes, False, ref_steps)
  our_lines = _run_main(monkeypatch, run_loop, episodes, True, our_steps)
  assert len(ref_lines) == N_EPISODES and ref_lines == our_lines
  assert len(ref_steps) == len(our_steps) > N_EPISODES
  for t, (a, b) in enumerate(zip(ref_steps, our_steps)):
    assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2], t
    assert np.array_equal(a[3], b[3]), t

"""Host-side behaviour added in round 2 (no GPU): static-label check of the lowering, the private RNG state of
the sampler's label probes, the content-hash build cache, bench.py's bookkeeping."""
import importlib.util
import os

import numpy as np
import pytest

from spriteworld_amd import action_spaces, build, device_sampler, lowering, renderers, tasks
from spriteworld_amd import factor_distributions as distribs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
  spec = importlib.util.spec_from_file_location('bench', os.path.join(ROOT, 'bench.py'))
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  return mod


@pytest.mark.parametrize('key', ['x', 'y'])
def test_filters_on_moving_factors_are_refused(key):
  """tasks.py:134-137,196-204 evaluate contains(sprite.factors) every step; the pool label is per episode, so
  a filter / cluster distribution that keys on x or y cannot be lowered."""
  rend = {'image': renderers.PILRenderer(image_size=(64, 64))}
  moving = distribs.Continuous(key, 0., 0.5)
  with pytest.raises(lowering.LoweringError):
    lowering.lower_config(tasks.FindGoalPosition(filter_distrib=moving), action_spaces.SelectMove(), rend)
  with pytest.raises(lowering.LoweringError):
    lowering.lower_config(tasks.Clustering([moving, distribs.Continuous('c0', 0., 0.5)]), action_spaces.SelectMove(), rend)
  meta = tasks.MetaAggregated([tasks.FindGoalPosition(filter_distrib=distribs.Continuous('c0', 0., 0.5)),
                               tasks.FindGoalPosition(filter_distrib=moving)])
  with pytest.raises(lowering.LoweringError):
    lowering.lower_config(meta, action_spaces.SelectMove(), rend)
  # static keys are fine
  lowering.lower_config(tasks.FindGoalPosition(filter_distrib=distribs.Continuous('c0', 0., 0.5)), action_spaces.SelectMove(), rend)


def test_label_probes_leave_the_global_numpy_stream_alone():
  factors = distribs.Product([distribs.Continuous('x', 0.1, 0.9), distribs.Continuous('y', 0.1, 0.9),
                              distribs.Discrete('shape', ['square']), distribs.Discrete('scale', [0.13]),
                              distribs.Continuous('c0', 0.0, 0.4), distribs.Continuous('c1', 0.3, 1.),
                              distribs.Continuous('c2', 0.9, 1.)])
  sampler = device_sampler.DeviceSampler([(factors, 3)], shuffle=True)
  task = tasks.FindGoalPosition(filter_distrib=distribs.Continuous('c0', 0., 0.4))
  rend = {'image': renderers.PILRenderer(image_size=(64, 64), color_to_rgb=renderers.hsv_to_rgb)}
  np.random.seed(123)
  want = np.random.uniform(size=4)
  np.random.seed(123)
  sampler.lower(task, rend)
  assert np.array_equal(np.random.uniform(size=4), want)


def test_build_cache_is_keyed_by_content():
  out = build.build()
  assert os.path.exists(out) and build.built_hash() == build.source_hash()
  before = os.path.getmtime(out)
  assert build.build() == out and os.path.getmtime(out) == before        # same sources and flags: reused
  assert len(build.source_hash()) == 64


def test_bench_bookkeeping():
  bench = _bench()
  from spriteworld_amd import workloads
  cfg, _, _ = workloads.build('cluster_s5', 4, 1)
  assert bench.algorithmic_bytes(cfg) == 12461
  cfg, _, _ = workloads.build('embodied_s12', 4, 1)
  assert bench.algorithmic_bytes(cfg) == 49513
  assert bench.profiled_counters('cluster_s5', 8192, 5, 'not-a-build-id') is None   # stale figures are never reported
  assert bench.usable_cores() >= 1

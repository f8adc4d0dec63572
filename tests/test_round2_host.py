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
def test_filters_on_moving_factors_lower_to_position_cuts(key):
  """tasks.py:134-137,196-204 evaluate contains(sprite.factors) every step.  Until round 5 a filter / cluster distribution that
  keys on x or y was refused; now its interval bounds become the task's position cuts (swb_task::xcuts / ycuts) and the pool
  tabulates every sprite's label per cell.  What still cannot be lowered says so: more than SWB_MAX_CUTS bounds on an axis,
  membership by EQUALITY with a position (Discrete), and sampling such a task's episodes on the device."""
  rend = {'image': renderers.PILRenderer(image_size=(64, 64))}
  moving = distribs.Continuous(key, 0., 0.5)
  cfg = lowering.lower_config(tasks.FindGoalPosition(filter_distrib=moving), action_spaces.SelectMove(), rend)
  t = cfg.tasks[0]
  n, cuts = (t.n_xcuts, t.xcuts) if key == 'x' else (t.n_ycuts, t.ycuts)
  assert (n, cuts[0], cuts[1]) == (2, 0.0, 0.5) and t.n_xcuts + t.n_ycuts == 2
  cfg = lowering.lower_config(tasks.Clustering([moving, distribs.Continuous('c0', 0., 0.5)]), action_spaces.SelectMove(), rend)
  assert cfg.tasks[0].n_xcuts + cfg.tasks[0].n_ycuts == 2
  meta = tasks.MetaAggregated([tasks.FindGoalPosition(filter_distrib=distribs.Continuous('c0', 0., 0.5)),
                               tasks.FindGoalPosition(filter_distrib=moving)])
  cfg = lowering.lower_config(meta, action_spaces.SelectMove(), rend)
  assert cfg.tasks[0].n_xcuts + cfg.tasks[0].n_ycuts == 0 and cfg.tasks[1].n_xcuts + cfg.tasks[1].n_ycuts == 2
  # float32 positions compare with the bound rounded to float32 (NEP 50); float64 positions with the bound itself
  cfg32 = lowering.lower_config(tasks.FindGoalPosition(filter_distrib=distribs.Continuous(key, 0.1, 0.7)), action_spaces.SelectMove(), rend,
                                pos_is_f32=True)
  cfg64 = lowering.lower_config(tasks.FindGoalPosition(filter_distrib=distribs.Continuous(key, 0.1, 0.7)), action_spaces.SelectMove(), rend,
                                pos_is_f32=False)
  c32 = cfg32.tasks[0].xcuts if key == 'x' else cfg32.tasks[0].ycuts
  c64 = cfg64.tasks[0].xcuts if key == 'x' else cfg64.tasks[0].ycuts
  assert (c32[0], c32[1]) == (float(np.float32(0.1)), float(np.float32(0.7))) and (c64[0], c64[1]) == (0.1, 0.7)
  # static keys: no cuts
  cfg = lowering.lower_config(tasks.FindGoalPosition(filter_distrib=distribs.Continuous('c0', 0., 0.5)), action_spaces.SelectMove(), rend)
  assert cfg.tasks[0].n_xcuts + cfg.tasks[0].n_ycuts == 0
  with pytest.raises(lowering.LoweringError):      # six bounds on one axis
    many = distribs.Mixture([distribs.Continuous(key, 0.1 * k, 0.1 * k + 0.05) for k in range(3)])
    lowering.lower_config(tasks.FindGoalPosition(filter_distrib=many), action_spaces.SelectMove(), rend)
  with pytest.raises(lowering.LoweringError):      # equality with a position
    lowering.lower_config(tasks.FindGoalPosition(filter_distrib=distribs.Discrete(key, [0.5])), action_spaces.SelectMove(), rend)
  factors = distribs.Product([distribs.Continuous('x', 0.1, 0.9), distribs.Continuous('y', 0.1, 0.9), distribs.Discrete('shape', ['square']),
                              distribs.Discrete('scale', [0.13]), distribs.Continuous('c0', 0.0, 0.4), distribs.Continuous('c1', 0.3, 1.),
                              distribs.Continuous('c2', 0.9, 1.)])
  with pytest.raises(lowering.LoweringError):      # episodes of such a task are drawn on the host
    device_sampler.DeviceSampler([(factors, 3)]).lower(tasks.FindGoalPosition(filter_distrib=moving), rend)


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
  # stale figures are never reported -- and the caller is told that they ARE stale (not merely absent)
  assert bench.profiled_counters('cluster_s5', 8192, 5, 'not-a-build-id') == (None, 'stale')
  assert bench.profiled_counters('no_such_workload', 8192, 5, 'not-a-build-id') == (None, 'none')
  assert bench.usable_cores() >= 1


def test_bench_line_carries_the_committed_counters_of_this_build():
  """bench.assemble_line with the figures of a finished run: for the shipped build (content hash of the kernel
  sources) the committed PMC passes fill `roofline.traffic` / `roofline.instructions`; the line is valid JSON with
  the contract's keys; another build id gets nulls."""
  import argparse
  import json
  bench = _bench()
  build_id = build.source_hash()[:16]
  for workload, aa, cover, kernel, image, sprites, a_bytes in (
      ('cluster_s5', 5, 'swb_cover_kernel<10>', 'swb_resample_kernel<6>', [64, 64], 5, 12461),
      ('cluster_s5', 1, 'swb_cover_kernel<2>', 'none (the cover kernel paints the frame)', [64, 64], 5, 12461),
      ('embodied_s12', 5, 'swb_cover_kernel<20>', 'swb_resample_kernel<6>', [128, 128], 12, 49513)):
    args = argparse.Namespace(envs_per_gpu=8192, gpus=1, steps=20, warmup=5, workload=workload, aa=aa)
    res = dict(elapsed=0.0046, kernel_ms=4.4, cover_ms=1.9, resample_ms=2.5, launches=20, a_bytes=a_bytes, errors=0,
               variant=dict(kernel=kernel, cover_kernel=cover, build_id=build_id, lds_bytes_per_wave=7200, waves_per_simd=5,
                            resample_waves_per_simd=8, n_bands=1, n_column_groups=1, nw=10, ncol=1, vs=6),
               facts=dict(sprites=sprites, image=image, anti_aliasing=aa, action_space='SelectMove', task='Clustering',
                          max_episode_length=50))
    line = json.loads(json.dumps(bench.assemble_line(args, res, 0.0046)))
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                'vs_baseline', 'dtype', 'data', 'config', 'roofline'):
      assert key in line, key
    r = line['roofline']
    assert r['bound'] == 'hbm' and r['peak'] == 8000.0 and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12
    one_kernel = kernel.startswith('none')          # anti_aliasing = 1, narrow image: the cover kernel is the whole step
    assert r['kernel'] == (cover + ' (paints the frame: anti_aliasing = 1)' if one_kernel else cover + ' + ' + kernel)
    assert r['build_id'] == build_id
    # the step's rate is taken over BOTH kernels; each kernel's own duration is listed, and they add up
    assert [k['name'] for k in r['kernels']] == ([cover] if one_kernel else [cover, kernel])
    assert abs(sum(k['ms'] for k in r['kernels']) - r['kernel_ms']) < 1e-9
    assert abs(r['achieved'] - a_bytes * 8192 / (r['kernel_ms'] * 1e-3) / 1e9) < 1e-6
    counters, state = bench.profiled_counters(workload, 8192, aa, build_id)
    assert r['counters'] == state
    if counters is None:        # the kernel sources changed after the last PMC passes: stale figures are not reported,
      assert state == 'stale'   # and the line says so
      assert r['traffic'] is None and r['instructions'] is None and r['bound_measured'] is None
      continue
    assert state == 'fresh' and r['bound_measured'] is not None
    assert counters['kernel'] == r['kernel']
    assert r['traffic'] == counters['hbm_traffic_bytes_per_launch'] > a_bytes * 8192
    ins = r['instructions']
    assert ins['insts_valu_per_env'] > 1000 and ins['valu_issue_frac_of_step'] > 0.2
    if workload == 'cluster_s5' and aa == 5:        # (the made-up kernel time above is about the headline's)
      assert ins['valu_issue_frac_of_step'] < 1.0
    if ins['resample_valu_model_min_per_env']:
      # (the model counts 10 per further span and 17 per finished row; the round-6 kernel issues 9 and 16, so a scene of many
      # spans per run -- 12 sprites at 128x128 -- measures a few per cent BELOW the model)
      assert 0.9 <= ins['resample_valu_measured_over_model'] < 2.0
    res['variant'] = dict(res['variant'], build_id='0000000000000000')
    stale = bench.assemble_line(args, res, 0.0046)['roofline']
    assert stale['traffic'] is None and stale['instructions'] is None and stale['counters'] == 'stale'

"""The kernel SOURCE against the oracle and the reference's recorded outputs, on the CPU.

tests/emu compiles spriteworld_amd/csrc (the C-ABI host side and the fused step kernel, every variant incl. the OV
builds) for the host against an emulation of the HIP runtime and of the wave-level builtins: work-items are fibres,
cross-lane operations (ballot, readlane, shuffles, DPP moves, wave barriers) are rendezvous of the 64 lanes that also
check that all lanes arrive from the same source line.  These tests are what the CPU suite can say about the device
code where no GPU exists: same inputs, same bar as the `-m gpu` parity tests (state, rewards, step types, discounts
bit-exact; frames +-0), at sizes the emulator finishes in seconds.

TEST INFRASTRUCTURE: the emulated library is never loaded by the product, proves nothing about timing, register
allocation or anything else the hipcc build adds -- the `-m gpu` tests run the real thing.  What it does prove is the
arithmetic and control flow of the kernel source, e.g. before a kernel change is taken to the (scarce) GPU.
"""
import numpy as np
import pytest

from spriteworld_amd import workloads
from tests import _util


def _bits(a):
  return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def _emu(cfg, pool):
  from tests import _emu_engine
  return _emu_engine.EmuEngine(cfg, pool)


def _emu_torch(cfg, pool):
  from tests import _emu_engine
  return _emu_engine.EmuTorchEngine(cfg, pool)


def _run(name, n_envs, steps, aa, seed=0):
  from oracle import oracle
  cfg, pool, sample = workloads.build(name, n_envs, episodes_per_env=3, seed=seed, anti_aliasing=aa)
  ora, eng = oracle.Engine(cfg, pool), _emu(cfg, pool)
  rng = np.random.default_rng(seed + 100)
  for t in range(steps):
    a = sample(rng)
    want = ora.step(a)
    eng.step(a)
    got = eng.outputs_host()
    st_o, st_g = ora.state(), eng.state()
    assert not got['error'].any(), (t, np.flatnonzero(got['error'])[:8])
    np.testing.assert_array_equal(got['step_type'], want['step_type'], err_msg='step_type t=%d' % t)
    np.testing.assert_array_equal(_bits(st_g['x']), _bits(st_o['x']), err_msg='x t=%d' % t)
    np.testing.assert_array_equal(_bits(st_g['y']), _bits(st_o['y']), err_msg='y t=%d' % t)
    for k in ('step_count', 'reset_next', 'episode', 'pool_entry', 'n_sprites'):
      np.testing.assert_array_equal(st_g[k], st_o[k], err_msg='%s t=%d' % (k, t))
    np.testing.assert_array_equal(got['success'], want['success'], err_msg='success t=%d' % t)
    np.testing.assert_array_equal(got['discount'].view(np.uint32), want['discount'].view(np.uint32))
    gr, wr = got['reward'], want['reward']
    assert np.array_equal(np.isnan(gr), np.isnan(wr)), 'reward NaN pattern t=%d' % t
    ok = ~np.isnan(wr)
    np.testing.assert_array_equal(_bits(gr[ok]), _bits(wr[ok]), err_msg='reward t=%d' % t)
    diff = np.abs(got['obs'].astype(np.int16) - want['obs'].astype(np.int16))
    assert diff.max() == 0, ('frame diff', int(diff.max()), int((diff > 0).sum()), t, np.argwhere(diff > 0)[:5].tolist())
  eng.close()


# the workloads of tests/test_gpu_parity.py (every kernel variant, every task / action space / dtype), a few environments each
@pytest.mark.parametrize('name,n_envs,steps,aa', [
    ('goal_s5', 3, 4, 5), ('cluster_s5', 3, 4, 5), ('goal_s5', 3, 3, 1), ('cluster_s5', 3, 3, 1), ('embodied_s12', 2, 3, 5),
    ('sorting_s4', 3, 4, 5), ('f64_drag', 4, 5, 3), ('f64_cluster', 4, 5, 3), ('cluster6_s12', 3, 4, 2), ('cluster9_s16', 3, 4, 2), ('ragged_s16', 8, 4, 5),
    ('ragged_s16_embodied', 8, 4, 5), ('wide_s4', 2, 3, 5), ('wide_s4', 2, 3, 1), ('tiny_s6', 4, 3, 5), ('tiny_s6', 4, 3, 1),
    ('goal_s5_f32a', 3, 4, 5), ('cluster_s5_f32a', 3, 4, 5), ('f64_drag_f32a', 3, 4, 3), ('f64_cluster_f32a', 3, 4, 3),
    ('sorting_s4_f32a', 3, 4, 5)])
def test_emulated_kernel_equals_oracle(name, n_envs, steps, aa):
  _run(name, n_envs, steps, aa)


@pytest.mark.parametrize('n_vertices', [33, 40, 64])
@pytest.mark.parametrize('name,n_envs,aa', [('cluster_s5', 3, 5), ('goal_s5', 3, 1), ('tiny_s6', 4, 5)])
def test_emulated_kernel_shapes_of_33_to_64_edges(n_vertices, name, n_envs, aa):
  """The C ABI takes shapes of up to SWB_MAX_SHAPE_VERTS = 64 vertices; from 33 on an edge gets ONE lane of the edge-lane scatter
  (G = 64 / n_edges = 1), whose lane -> edge reciprocal is then 2^16 itself (round 4 packed it in 16 bits: every lane served
  edge 0 -- ADVICE round 4).  No built-in shape has more than 30 vertices: the circle is swapped for a regular n-gon."""
  from spriteworld_amd import shapes
  with _util.swapped_shape('circle', shapes.polygon(n_vertices)):
    _run(name, n_envs, 3, aa)


@pytest.mark.parametrize('geom,aa', [('96x48', 3), ('48x96', 2), ('256x64', 2), ('160x160', 4), ('128x128', 1),
                                     ('100x60', 3), ('64x256', 1), ('32x32', 8), ('32x512', 4)])
def test_emulated_kernel_image_geometries(geom, aa):
  _run('geom_' + geom, 2, 2, aa)


@pytest.mark.parametrize('seed', range(8))
def test_emulated_kernel_randomised_configurations(seed):
  _run('fuzz_%d' % seed, 3, 3, 5, seed=seed)


def test_emulated_run_list_capacity_follows_the_sprite_count():
  """Ten sprites on a 60-row canvas (192 x 60 at anti_aliasing = 3; seed 2681 of tools/fuzz_sweep.py, found in round 5): rows of
  four to eight visible spans, 241 .. 300 units of run list for 60 rows -- the capacity was 4 units per canvas row whatever the
  sprite count, the environment was flagged SWB_ENV_ERR_SPAN_OVERFLOW at step 7 and its frame was short of a batch of rows.
  Now (S + 1) units per row: any scene of convex sprites, folded or not."""
  _run('fuzz_2681', 64, 8, 5, seed=2681)


@pytest.mark.parametrize('name,n_envs,aa', [('embodied_s12', 2, 5), ('ragged_s16', 6, 5), ('cluster_s5', 3, 1)])
def test_emulated_kernel_span_overflow_slots(monkeypatch, name, n_envs, aa):
  """Rows with more than three visible spans take the HBM overflow path when the LDS lists are switched off."""
  monkeypatch.setenv('SWB_LDS_SPAN_CAP', '0')
  _run(name, n_envs, 3, aa)


# ---- the hand-off between the two kernels of a step (round 3): bands of output rows, cost-ordered dispatch, list capacity
@pytest.mark.parametrize('bands,band_tasks', [(1, 0), (2, 0), (2, 1), (3, 0), (3, 1), (8, 0), (8, 1)])
@pytest.mark.parametrize('name,n_envs,aa', [('cluster_s5', 3, 5), ('embodied_s12', 2, 5), ('geom_100x60', 2, 3), ('geom_64x256', 2, 1),
                                            ('geom_96x48', 2, 3), ('cluster_s5', 3, 1)])
def test_emulated_kernel_any_number_of_bands(monkeypatch, bands, name, n_envs, aa, band_tasks):
  """The resample / fill kernel splits an image into bands of output rows (one wave each); a band starts with the output rows
  already in flight at its first canvas row.  Every band count gives the same frames -- whether the second kernel's tasks are
  whole lists (every band of a list in the list's place of the cost order) or single bands filed under their own cost
  (round 6, swb_params::band_tasks: what launches of four waves or more per SIMD use)."""
  monkeypatch.setenv('SWB_BANDS', str(bands))
  monkeypatch.setenv('SWB_BAND_TASKS', str(band_tasks))
  _run(name, n_envs, 3, aa)


def test_emulated_kernel_without_cost_ordered_dispatch(monkeypatch):
  monkeypatch.setenv('SWB_NO_COST_ORDER', '1')
  _run('cluster_s5', 11, 4, 5)
  _run('cluster_s5', 11, 3, 1)
  _run('geom_256x64', 3, 2, 2)


@pytest.mark.parametrize('name,n_envs,aa', [('cluster_s5', 19, 5), ('embodied_s12', 5, 5), ('ragged_s16', 13, 5), ('geom_128x128', 9, 1),
                                            ('sorting_s4', 11, 5), ('cluster_s5', 13, 1), ('tiny_s6', 7, 1)])
def test_emulated_cover_launches_in_cost_order(monkeypatch, name, n_envs, aa):
  """Launches of more than one round of cover waves take the environments in order of what their cover wave cost in the previous
  launch (cycle counts filed per environment, heavy scenes first); SWB_COVER_ORDER asks for it at any batch size.  The order is
  only used after a launch that filed every environment: a step without an observation in between falls back to the plain order
  for one launch.  State, rewards and frames do not depend on any of it."""
  from oracle import oracle
  monkeypatch.setenv('SWB_COVER_ORDER', '1')
  _run(name, n_envs, 4, aa)
  cfg, pool, sample = workloads.build(name, n_envs, episodes_per_env=3, seed=1, anti_aliasing=aa)
  ora, eng = oracle.Engine(cfg, pool), _emu(cfg, pool)
  rng = np.random.default_rng(7)
  for t in range(6):
    a = sample(rng)
    want = ora.step(a)
    eng.step(a, render=(t != 2))                     # launch 2 renders nothing and files nothing
    if t == 2:
      continue
    got = eng.outputs_host()
    np.testing.assert_array_equal(got['step_type'], want['step_type'])
    np.testing.assert_array_equal(_bits(eng.state()['x']), _bits(ora.state()['x']))
    assert np.array_equal(got['obs'], want['obs']), t
  eng.close()


@pytest.mark.parametrize('seed', [1, 4, 6])
def test_emulated_cover_cost_order_on_randomised_configurations(monkeypatch, seed):
  """Random task / action space / geometry / sprite count configurations (the fuzz workloads) with the cost-ordered cover launch
  and short dealing rounds switched on: the dispatch never changes a result."""
  monkeypatch.setenv('SWB_COVER_ORDER', '1')
  monkeypatch.setenv('SWB_DEAL_SHIFT', '1')
  _run('fuzz_%d' % seed, 21, 4, 5, seed=seed)


@pytest.mark.parametrize('shift', ['1', '2'])
def test_emulated_kernel_cost_order_dealt_in_alternating_rounds(monkeypatch, shift):
  """The resample / fill blocks of a shard take its cost-ordered tasks in rounds of 2^deal_shift blocks, odd rounds in ascending
  order (32 blocks per round on the GPU: the compute units of an XCD; SWB_DEAL_SHIFT shortens the rounds so that small batches
  have several, with a short last one).  Every task is served exactly once."""
  monkeypatch.setenv('SWB_DEAL_SHIFT', shift)
  for n in (33, 70, 97):
    _run('cluster_s5', n, 2, 5)
  _run('geom_256x64', 21, 2, 2)
  _run('geom_128x128', 35, 2, 1)


@pytest.mark.parametrize('name,aa', [('embodied_s12', 5), ('geom_256x64', 2), ('geom_128x128', 1)])
def test_emulated_kernel_cost_order_files_every_column_group(name, aa):
  """Images wider than 64 columns: every (environment, group of 64 columns) is a task of its own in the cost-ordered lists,
  filed under the length of that group's run list."""
  for n in (1, 7, 9):
    _run(name, n, 2, aa)


@pytest.mark.parametrize('name,aa', [('cluster_s5', 5), ('cluster_s5', 1)])
def test_emulated_kernel_cost_order_covers_every_environment(name, aa):
  """Sizes that do not divide by the eight shards or the four waves of a resample block."""
  for n in (1, 7, 9, 33):
    _run(name, n, 3, aa)


@pytest.mark.parametrize('name,n_envs', [('cluster_s5', 3), ('tiny_s6', 4), ('wide_s4', 2), ('ragged_s16', 8)])
def test_emulated_fill_kernel_for_narrow_images(monkeypatch, name, n_envs):
  """anti_aliasing = 1: images of up to 64 columns are painted by the cover kernel itself (the default, in every other AA = 1
  case of this file); SWB_NO_PAINT_IN_COVER sends them through the run lists and the fill kernel like wider images."""
  monkeypatch.setenv('SWB_NO_PAINT_IN_COVER', '1')
  _run(name, n_envs, 3, 1)


@pytest.mark.parametrize('run_cap,bands,n_envs', [(24, 1, 4), (8, 4, 1), (12, 8, 3), (40, 2, 5)])
def test_emulated_kernel_run_list_overflow_is_flagged(monkeypatch, run_cap, bands, n_envs):
  """A run list that does not fit its capacity (swb_params::run_cap; SWB_RUN_CAP lowers it) flags the environment
  (SWB_ENV_ERR_SPAN_OVERFLOW) instead of writing past it -- and the second kernel never READS past it either: every band
  of an overflowed list starts inside the written part (round-3 advice: with SWB_RUN_CAP=8, SWB_BANDS=4 a band header
  pointed 114 units beyond an 8-unit list).  The emulated build counts run-record reads at or beyond the capacity."""
  import ctypes as C
  from spriteworld_amd import _abi
  monkeypatch.setenv('SWB_RUN_CAP', str(run_cap))
  monkeypatch.setenv('SWB_ARENA_UNITS', '0')               # (no shared arena to continue in: the round-5 layout)
  monkeypatch.setenv('SWB_BANDS', str(bands))
  monkeypatch.setenv('SWB_BAND_TASKS', str(n_envs & 1))    # (either form of the second kernel's tasks)
  for aa, name in ((5, 'cluster_s5'), (1, 'wide_s4')):
    if aa == 1:
      monkeypatch.setenv('SWB_NO_PAINT_IN_COVER', '1')
    cfg, pool, sample = workloads.build(name, n_envs, episodes_per_env=2, seed=0, anti_aliasing=aa)
    eng = _emu(cfg, pool)
    eng.lib.emu_violations.restype = C.c_long
    eng.lib.emu_violations(1)
    rng = np.random.default_rng(0)
    for _ in range(3):
      eng.step(sample(rng))
    got = eng.outputs_host()
    assert (got['error'] & _abi.ENV_ERR_SPAN_OVERFLOW).any()
    assert eng.lib.emu_violations(1) == 0
    eng.close()


@pytest.mark.parametrize('run_cap,bands,arena', [(8, 1, 1 << 20), (8, 4, 1 << 20), (12, 8, 1 << 20), (40, 2, 1 << 20), (24, 1, 600)])
def test_emulated_kernel_run_lists_continue_in_the_shared_arena(monkeypatch, run_cap, bands, arena):
  """Round 6: a run list owns a part of its own and MOVES to a segment of a shared arena, twice (four times ...) as large, when
  it outgrows it (the wave copies what it wrote; positions in the header are counted from the own part, so the second kernels
  know nothing of it).  With an own part of 8 .. 40 units EVERY list moves, several times: frames, state and rewards stay
  bit-exact on both second kernels (resample: anti_aliasing 5; fill: anti_aliasing 1 on a wide image) for every band count, no
  environment is flagged, and no run record is read outside the list's own part or the arena.  With an arena too small for
  the batch (600 units) the environments that find it exhausted are FLAGGED and every other one is still exact."""
  import ctypes as C
  from oracle import oracle
  from spriteworld_amd import _abi
  monkeypatch.setenv('SWB_RUN_CAP', str(run_cap))
  monkeypatch.setenv('SWB_ARENA_UNITS', str(arena))
  monkeypatch.setenv('SWB_BANDS', str(bands))
  monkeypatch.setenv('SWB_BAND_TASKS', '1')                # (a moving list shifts the band starts it has recorded -- and their copy in LDS)
  for aa, name, n_envs in ((5, 'cluster_s5', 5), (1, 'geom_160x48', 3), (5, 'embodied_s12', 2)):
    if aa == 1:
      monkeypatch.setenv('SWB_NO_PAINT_IN_COVER', '1')
    cfg, pool, sample = workloads.build(name, n_envs, episodes_per_env=2, seed=1, anti_aliasing=aa)
    eng, ora = _emu(cfg, pool), oracle.Engine(cfg, pool)
    eng.lib.emu_violations.restype = C.c_long
    eng.lib.emu_violations(1)
    rng = np.random.default_rng(0)
    flagged_any = False
    for _ in range(3):
      a = sample(rng)
      want = ora.step(a)
      eng.step(a)
      got = eng.outputs_host()
      flagged = (got['error'] & _abi.ENV_ERR_SPAN_OVERFLOW) != 0
      flagged_any |= bool(flagged.any())
      if arena >= (1 << 20):
        assert not flagged.any()
      assert np.array_equal(got['obs'][~flagged], want['obs'][~flagged])
      assert np.array_equal(got['step_type'], want['step_type'])
      ok = ~np.isnan(want['reward'])
      assert np.array_equal(got['reward'][ok].view(np.uint64), want['reward'][ok].view(np.uint64))
    if arena < (1 << 20) and name == 'embodied_s12':
      assert flagged_any                         # (two 12-sprite scenes at 128x128 need thousands of units: 600 are exhausted)
    assert eng.lib.emu_violations(1) == 0
    v = eng.variant()
    assert v['run_cap'] == run_cap and v['arena_units'] == arena and v['run_list_bytes'] > 0
    eng.close()


def test_emulated_run_lists_are_trimmed_after_the_third_rendering_launch():
  """The lists start with room for any scene of convex sprites (max(4, S + 1) units per canvas row); after the third rendering
  launch the engine cuts them to 1.25 x the longest list written + a shared arena (swb_trim_run_lists).  Frames stay exact
  before, at and after the cut; a list that later outgrows its part moves to the arena; a new pool restores the
  reservation."""
  from oracle import oracle
  for name, n_envs, aa in (('embodied_s12', 3, 5), ('cluster_s5', 6, 5), ('geom_160x48', 3, 1)):
    cfg, pool, sample = workloads.build(name, n_envs, episodes_per_env=3, seed=2, anti_aliasing=aa)
    eng, ora = _emu(cfg, pool), oracle.Engine(cfg, pool)
    rng = np.random.default_rng(5)
    sizes = []
    for t in range(7):
      a = sample(rng)
      want = ora.step(a)
      eng.step(a)
      got = eng.outputs_host()
      assert not got['error'].any()
      assert np.array_equal(got['obs'], want['obs']), (name, t)
      v = eng.variant()
      sizes.append((v['run_cap'], v['arena_units'], v['run_list_bytes']))
    worst = max(4, cfg.max_sprites + 1) * cfg.anti_aliasing * cfg.image_w + 1
    assert sizes[0][0] == sizes[1][0] == worst                     # the start-up reservation ...
    assert sizes[2][0] < worst // 2 and sizes[2][2] < sizes[1][2], sizes           # ... cut at the third launch
    fixed_before, fixed_after = sizes[1][2] - 8 * sizes[1][1], sizes[2][2] - 8 * sizes[2][1]
    assert fixed_after < fixed_before // 2, sizes                  # (the lists' own parts; the arena has a floor of eight worst-case lists)
    assert sizes[-1] == sizes[2]                                   # once
    assert sizes[2][1] >= 16 * worst                               # the arena: at least sixteen worst-case lists
    assert eng.trim() == sizes[2][0]                               # (calling it again changes nothing)
    eng.set_pool(pool)                                             # a new pool: the full reservation again
    eng.step(sample(rng))
    assert eng.variant()['run_cap'] == worst
    eng.close()


@pytest.mark.parametrize('name', _util.golden_cases())
def test_emulated_kernel_reproduces_the_reference_fixtures(name):
  """tests/golden/*.npz (outputs recorded from the unmodified reference): the first 50 steps of every fixture."""
  cfg, pool, z = _util.load_golden(name)
  eng = _emu(cfg, pool)

  def step(a):
    eng.step(a)
    return eng.outputs_host()

  short = {k: z[k] for k in z.files}
  short['actions'] = z['actions'][:50]
  _util.check_against_golden(eng, cfg, short, eng.state, step, 'emu/' + name)
  eng.close()


@pytest.mark.parametrize('name,n_envs,steps,aa', [('goal_s5', 6, 5, 5), ('cluster_s5', 6, 4, 1), ('embodied_s12', 3, 3, 5),
                                                   ('geom_256x64', 3, 3, 2), ('geom_32x32', 3, 3, 8)])
def test_emulated_ov_kernels_sprite_setters(name, n_envs, steps, aa):
  """The scenarios of tests/test_gpu_setters.py on the emulated library: swb_set_sprite_attr's host arithmetic, the
  override arrays and the OV builds of the step kernel against the oracle's setters."""
  from tests import _setter_cases
  _setter_cases.run_parity(_emu_torch, name, n_envs, steps, aa)


def test_emulated_setters_factors_and_reset():
  from tests import _emu_engine, _setter_cases
  _setter_cases.factors_and_reset_case(_emu_torch, _emu_engine.EmuError)


def test_emulated_live_sprite_handles(monkeypatch):
  """environment.BatchedEnvironment + sprite.LiveSprite on the emulated library, checked against matplotlib."""
  from spriteworld_amd import environment
  from tests import _emu_engine, _setter_cases
  monkeypatch.setattr(environment._engine, 'Engine', _emu_engine.EmuTorchEngine)
  _setter_cases.live_sprite_case()


def test_emulator_refuses_cross_lane_operations_under_divergence():
  """The rendezvous checks the call site: the build carries a self-test kernel whose lanes reach two different
  ballots; it must be caught (subprocess: the emulator aborts)."""
  import subprocess
  import sys
  code = ('import ctypes, sys; sys.path.insert(0, %r); from tests.emu import build_emu; '
          'l = ctypes.CDLL(build_emu.build()); l.emu_selftest_divergent_ballot()' % _util.ROOT)
  p = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True)
  assert p.returncode != 0 and 'divergent control flow' in p.stderr, p.stderr[-500:]


# ---- the device-side reset sampler (swb_sampler.hip.inc, one work-item per pool entry) on the emulated library:
# the `-m gpu` tests of tests/test_device_sampler.py, called with the engine interface backed by the emulator
def _patch_engine(monkeypatch):
  from spriteworld_amd import environment
  from tests import _emu_engine
  monkeypatch.setattr(environment._engine, 'Engine', _emu_engine.EmuTorchEngine)


@pytest.mark.parametrize('case', ['cobra_like', 'embodied_like', 'holdouts', 'hsv_mixed', 'mixed_types', 'sorting_like'])
def test_emulated_sampler_kernel_equals_the_python_model(monkeypatch, case):
  from tests import test_device_sampler as T
  assert sorted(T.CASES) == ['cobra_like', 'embodied_like', 'holdouts', 'hsv_mixed', 'mixed_types', 'sorting_like']
  _patch_engine(monkeypatch)
  T.test_device_pool_matches_the_model_bit_for_bit(case)


def test_emulated_sampler_shards_and_refresh(monkeypatch):
  from tests import test_device_sampler as T
  _patch_engine(monkeypatch)
  T.test_shards_draw_the_episodes_of_the_whole_job()
  T.test_refresh_pool_redraws_everything_but_the_live_entries()


@pytest.mark.skipif(not __import__('oracle.ref_harness', fromlist=['x']).reference_available(), reason='reference tree not present')
def test_emulated_factors_kernel_equals_reference_sprite_factors():
  """renderers/handcrafted.py:29-82 SpriteFactors of the UNMODIFIED reference on a running environment (incl. sprites
  modified through the setters, sprite.py:152-175) against swb_factors of the emulated library, every step."""
  import copy
  from oracle import ref_harness
  ref_harness.load_reference()
  from spriteworld import action_spaces, environment, renderers, sprite, tasks
  from spriteworld_amd import _abi, lowering, shapes
  from spriteworld_amd import sprite as sprite_lib
  rng = np.random.RandomState(8)
  names = list(shapes.SHAPES.keys())

  def gen():
    return [sprite.Sprite(x=float(rng.uniform(0.2, 0.8)), y=float(rng.uniform(0.2, 0.8)), shape=str(rng.choice(names)),
                          angle=float(rng.choice([0, 40, 200])), scale=float(rng.choice([0.1, 0.2])),
                          c0=int(rng.randint(0, 256)), c1=int(rng.randint(0, 256)), c2=int(rng.randint(0, 256)),
                          x_vel=float(rng.uniform(-0.01, 0.01)), y_vel=float(rng.uniform(-0.01, 0.01)))
            for _ in range(int(rng.randint(1, 4)))]

  episodes = [gen() for _ in range(5)]
  task = tasks.FindGoalPosition(goal_position=(0.5, 0.5), terminate_distance=0.05)
  aspace = action_spaces.SelectMove(scale=0.2)
  rends = {'image': renderers.PILRenderer(image_size=(32, 32), anti_aliasing=2), 'factors': renderers.SpriteFactors()}
  cfg = lowering.lower_config(task, aspace, {'image': rends['image']}, True, 9, 1, 3,
                              pos_is_f32=(lowering.position_dtype(episodes) == np.float32))
  pool = lowering.lower_episodes(episodes, task, {'image': rends['image']}, max_sprites=3).assign_round_robin(1)
  eng = _emu(cfg, pool)
  it = (copy.deepcopy(e) for e in [episodes[0]] + episodes * 20)
  env = environment.Environment(task=task, action_space=aspace, renderers=rends, init_sprites=lambda: next(it),
                                keep_in_frame=True, max_episode_length=9)
  arng = np.random.RandomState(3)
  for t in range(40):
    if t % 4 == 2 and env._sprites and not env._reset_next_step:
      k = int(arng.randint(0, len(env._sprites)))
      attr, value = [('angle', 77.0), ('scale', 0.33), ('shape', 'star_5')][(t // 4) % 3]
      setattr(env._sprites[k], attr, value)
      eng.set_sprite_attr(0, k, {'shape': _abi.ATTR_SHAPE, 'angle': _abi.ATTR_ANGLE, 'scale': _abi.ATTR_SCALE}[attr],
                          shapes.shape_index(value) if attr == 'shape' else value)
    a = arng.uniform(0, 1, 4)
    ts = env.step(a)
    eng.step(a[None])
    got = eng.factors()[0]
    want = ts.observation['factors']
    assert len(want) == eng.state()['n_sprites'][0]
    for s, row in enumerate(want):
      assert [row[f] for f in sprite_lib.FACTOR_NAMES] == got[s].tolist(), (t, s)
    assert not got[len(want):].any()
    assert np.array_equal(ts.observation['image'], eng.outputs_host()['obs'][0]), t


def _reference_configs():
  from tests import test_oracle_vs_reference as R
  return R.CONFIGS


@pytest.mark.skipif(not __import__('oracle.ref_harness', fromlist=['x']).reference_available(), reason='reference tree not present')
@pytest.mark.parametrize('module,mode', _reference_configs())
def test_emulated_kernel_equals_the_unmodified_reference(module, mode):
  """Every shipped config in both modes (tests/configs/configs_test.py:33-58 runs the same grid): the UNMODIFIED reference
  `Environment` and the kernel source (emulated) stepped side by side in one process, no oracle in between -- step
  types, rewards, positions bit-exact, frames +-0, across resets."""
  import importlib
  from oracle import ref_harness
  ref_harness.load_reference()
  from spriteworld import environment
  from spriteworld import renderers as ref_renderers
  from spriteworld_amd import lowering
  from tests import test_oracle_vs_reference as R
  seed, n_eps, n_steps = 33, 12, 100
  np.random.seed(seed)
  config = importlib.import_module(module).get_config(mode)
  episodes = [config['init_sprites']() for _ in range(n_eps)]
  task, aspace, rends = config['task'], config['action_space'], config['renderers']
  S = max(len(e) for e in episodes)
  cfg = lowering.lower_config(task, aspace, rends, True, config['max_episode_length'], 1, S,
                              pos_is_f32=(lowering.position_dtype(episodes) == np.float32))
  pool = lowering.lower_episodes(episodes, task, rends, max_sprites=S).assign_round_robin(1)
  eng = _emu(cfg, pool)
  it = R._fresh_episodes(episodes)
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
    assert (np.isnan(r) and np.isnan(out['reward'][0])) or _bits(r) == _bits(out['reward'][0]), (t, r)
    assert bool(ts.observation['success']) == bool(out['success'][0]), t
    assert np.array_equal(ts.observation['image'], out['obs'][0]), t
    st = eng.state()
    pos = np.array([s.position for s in env._sprites], dtype=np.float64).reshape(-1, 2)
    n = st['n_sprites'][0]
    assert n == len(pos) and np.array_equal(pos[:, 0], st['x'][0, :n]) and np.array_equal(pos[:, 1], st['y'][0, :n]), t
  eng.close()


def test_emulated_results_do_not_depend_on_lane_order_or_lds_garbage():
  """The same parity tests with the lanes taking their turns in DESCENDING order between rendezvous and another garbage
  byte in the uninitialised LDS (both are read once per process: a subprocess).  A cross-lane hazard on LDS without a
  fence, or a read of LDS nothing wrote, would change a result."""
  import os
  import subprocess
  import sys
  env = dict(os.environ, SWB_EMU_LANE_ORDER='reverse', SWB_EMU_LDS_FILL='0x00')
  p = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-q', '-x', '-p', 'no:cacheprovider', '-k',
                      'equals_oracle and (cluster_s5 or embodied or f64_cluster or ragged_s16 or tiny or sorting) or setters and goal_s5'],
                     cwd=_util.ROOT, env=env, capture_output=True, text=True)
  assert p.returncode == 0 and ' passed' in p.stdout, p.stdout[-1500:]


# ---- host-API tests written for the GPU (tests/test_env_spec_conformance.py, test_gym_wrapper.py, test_host_api.py),
# run here with the engine interface backed by the emulated library: the dm_env / gym surface over the kernel source
def test_emulated_environment_conforms_to_its_specs(monkeypatch):
  """tests/environment_test.py:30-51 (dm_env EnvironmentTestMixin), as re-expressed for the N = 1 Environment."""
  from tests import test_env_spec_conformance as T
  _patch_engine(monkeypatch)
  for make in (T._reference_test_env, T._rendered_env):
    T.test_reset_and_step_protocol_on_fresh_environments(make)
    T.test_longer_action_sequence_conforms_to_the_specs(make)
  T.test_specs_are_specs()


def test_emulated_gym_wrapper_and_single_environment(monkeypatch):
  from tests import test_gym_wrapper as G
  from tests import test_host_api as H
  _patch_engine(monkeypatch)
  for embodied in (False, True):
    G.test_reference_gym_wrapper_episode_pattern(embodied)
  H.test_single_environment_follows_example_run_loop()
  H.test_sprite_factors_observation_and_action_noise()


@pytest.mark.parametrize('f32', [True, False], ids=['f32pos', 'f64pos'])
@pytest.mark.parametrize('name', __import__('tests._position_cases', fromlist=['CASES']).CASES)
def test_emulated_kernel_tasks_that_filter_on_position(name, f32):
  """Round 6: task filters / cluster distributions keyed on x, y (tests/_position_cases.py; pinned against the unmodified
  reference through the oracle in tests/test_oracle_vs_reference.py and through tests/golden/position_*.npz).  The kernel looks
  every sprite's label up in the cell of the task's position grid it stands in, every step."""
  from oracle import oracle
  from spriteworld_amd import lowering
  from tests import _position_cases as pc
  ns = pc.namespace_of_mirrors()
  task, aspace, rends, keep, max_len = pc.environment_parts(ns, name)
  n_envs = 5
  episodes = pc.episodes_of(ns, name, f32, n_episodes=3 * n_envs)
  cfg = lowering.lower_config(task, aspace, rends, keep, max_len, n_envs, pc.N_SPRITES, pos_is_f32=f32)
  pool = lowering.lower_episodes(episodes, task, rends, max_sprites=pc.N_SPRITES).assign_round_robin(n_envs, 3)
  assert pool.cell_label is not None
  ora, eng = oracle.Engine(cfg, pool), _emu(cfg, pool)
  rng = np.random.default_rng(11)
  flips = 0
  prev = None
  sticky = np.zeros(n_envs, np.uint8)
  for t in range(40):
    a = rng.uniform(0.0, 1.0, size=(n_envs, 4))
    st = ora.state()
    for i in range(0, n_envs, 2):                      # click ON a sprite in every second environment
      k = int(rng.integers(0, max(int(st['n_sprites'][i]), 1)))
      a[i, 0], a[i, 1] = st['x'][i, k], st['y'][i, k]
    want = ora.step(a)
    eng.step(a)
    got = eng.outputs_host()
    np.testing.assert_array_equal(got['step_type'], want['step_type'])
    np.testing.assert_array_equal(got['success'], want['success'])
    assert np.array_equal(np.isnan(got['reward']), np.isnan(want['reward']))
    ok = ~np.isnan(want['reward'])
    np.testing.assert_array_equal(_bits(got['reward'][ok]), _bits(want['reward'][ok]))
    sticky |= want['error']                            # (the engine's error flags are sticky; the oracle's are per step)
    np.testing.assert_array_equal(got['error'], sticky)
    np.testing.assert_array_equal(got['obs'], want['obs'])
    if prev is not None:
      flips += int((want['reward'] != prev).sum())
    prev = want['reward']
  assert flips > 20
  eng.close()

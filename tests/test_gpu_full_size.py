"""Full-size checks (BASELINE.json sizes): 8192 environments per GPU.

The oracle cannot step 8192 environments for long within a test budget, so the full batch is
checked through size-independent properties:
  * a random sample of the batch against the oracle stepping exactly those environments;
  * shard invariance: environments [k*B, (k+1)*B) run as their own engine produce the same bytes as
    the corresponding slice of the full run (this is the multi-GPU sharding of DESIGN.md section 5,
    exercised on one GPU);
  * determinism: two identical runs are byte-identical; a checksum of per-env frame checksums agrees.
"""
import zlib

import numpy as np
import pytest

from spriteworld_amd import lowering, workloads

pytestmark = pytest.mark.gpu

N = 8192


def _sub_pool(pool, envs, epe):
  """Pool restricted to the entries of `envs` (round-robin assignment with `epe` entries per env)."""
  idx = np.concatenate([np.arange(pool.pool_base[e], pool.pool_base[e] + pool.pool_len[e]) for e in envs])
  sub = lowering.Pool(len(idx), pool.max_sprites, pool.n_tasks)
  for f in lowering.Pool.FIELDS:
    if f in ('pool_base', 'pool_len'):
      continue
    setattr(sub, f, np.ascontiguousarray(getattr(pool, f)[idx]))
  sub.angle, sub.color = pool.angle[idx], pool.color[idx]
  sub.assign_round_robin(len(envs), epe)
  return sub


def _clone_cfg(cfg, n):
  import ctypes
  from spriteworld_amd import _abi
  c = _abi.SwbConfig.from_buffer_copy(bytes(cfg))
  c.n_envs = n
  return c


@pytest.mark.parametrize('name,n_envs,epe,steps', [('cluster_s5', N, 3, 24), ('goal_s5', N, 3, 24),
                                                   ('embodied_s12', N, 2, 12),       # BASELINE configs[4]
                                                   ('cluster_s5', 65536, 2, 8)])     # configs[3]'s batch in ONE launch
def test_sample_of_full_batch_matches_oracle(name, n_envs, epe, steps):
  import torch
  from oracle import oracle
  from spriteworld_amd import engine
  cfg, pool, sample = workloads.build(name, n_envs, episodes_per_env=epe, seed=5, anti_aliasing=5)
  eng = engine.Engine(cfg, pool)
  pick = np.sort(np.random.default_rng(1).choice(n_envs, 96, replace=False))
  if n_envs > N:                      # a big batch: only the sampled environments leave the device
    ora = oracle.Engine(_clone_cfg(cfg, len(pick)), _sub_pool(pool, pick, epe))
    rng = np.random.default_rng(9)
    idx = torch.as_tensor(pick, device=eng.device)
    for t in range(steps):
      a = sample(rng)
      eng.step(a)
      want = ora.step(a[pick])
      assert int(eng.error.max().item()) == 0
      assert np.array_equal(eng.obs[idx].cpu().numpy(), want['obs']), t
      assert np.array_equal(eng.step_type[idx].cpu().numpy(), want['step_type']), t
      assert np.array_equal(eng.reward[idx].cpu().numpy().view(np.uint64), want['reward'].view(np.uint64)), t
      assert np.array_equal(eng.success[idx].cpu().numpy(), want['success']), t
    st_g, st_o = eng.state(), ora.state()
    assert np.array_equal(st_g['x'][pick], st_o['x']) and np.array_equal(st_g['y'][pick], st_o['y'])
    eng.close()
    return
  ora = oracle.Engine(_clone_cfg(cfg, len(pick)), _sub_pool(pool, pick, epe))
  rng = np.random.default_rng(9)
  for t in range(steps):
    a = sample(rng)
    eng.step(a)
    got = eng.outputs_host()
    want = ora.step(a[pick])
    assert not got['error'].any()
    assert np.array_equal(got['obs'][pick], want['obs']), t
    assert np.array_equal(got['step_type'][pick], want['step_type']), t
    assert np.array_equal(got['reward'][pick].view(np.uint64), want['reward'].view(np.uint64)), t
    st_g, st_o = eng.state(), ora.state()
    assert np.array_equal(st_g['x'][pick], st_o['x']) and np.array_equal(st_g['y'][pick], st_o['y']), t
  eng.close()


def test_shards_equal_slices_of_the_full_run_and_runs_are_deterministic():
  from spriteworld_amd import engine
  epe, steps, shards = 2, 12, 8
  cfg, pool, sample = workloads.build('cluster_s5', N, episodes_per_env=epe, seed=6, anti_aliasing=5)
  rng = np.random.default_rng(3)
  actions = [sample(rng) for _ in range(steps)]

  def run(cfg_, pool_, sl):
    e = engine.Engine(cfg_, pool_)
    frames, rewards = [], []
    for a in actions:
      e.step(a[sl])
      o = e.outputs_host()
      frames.append(o['obs'])
      rewards.append(o['reward'])
    e.close()
    return frames, rewards

  full_f, full_r = run(cfg, pool, slice(None))
  again_f, again_r = run(cfg, pool, slice(None))
  crc = lambda fs: zlib.crc32(np.array([zlib.crc32(f[i].tobytes()) for f in fs for i in range(0, N, 7)], np.uint32).tobytes())
  assert crc(full_f) == crc(again_f)
  assert all(np.array_equal(a.view(np.uint64), b.view(np.uint64)) for a, b in zip(full_r, again_r))
  B = N // shards
  for k in (0, 3, shards - 1):
    envs = np.arange(k * B, (k + 1) * B)
    sh_f, sh_r = run(_clone_cfg(cfg, B), _sub_pool(pool, envs, epe), slice(k * B, (k + 1) * B))
    for t in range(steps):
      assert np.array_equal(sh_f[t], full_f[t][k * B:(k + 1) * B]), (k, t)
      assert np.array_equal(sh_r[t].view(np.uint64), full_r[t][k * B:(k + 1) * B].view(np.uint64)), (k, t)


def test_bench_in_run_verification_on_the_engine():
  """bench.py's own check of a timed run (gpu_run(verify=...) + verify_against_oracle), on the HIP engine: 0 mismatches over
  warm-up + timed steps, and a JSON-serialisable block."""
  import importlib.util
  import json
  import os
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  spec = importlib.util.spec_from_file_location('bench', os.path.join(root, 'bench.py'))
  bench = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(bench)
  for name, aa in (('cluster_s5', 5), ('embodied_s12', 5), ('cluster_s5', 1)):
    res = bench.gpu_run(name, 512, steps=21, warmup=4, aa=aa, device=0, verify=48)
    assert res['error'] is None and res['errors'] == 0
    out = bench.verify_against_oracle(res['sample'])
    assert out['verified_envs'] == 48 and out['mismatches'] == 0 and out['frame_bytes_differing'] == 0, (name, aa, out)
    assert out['steps_replayed'] == 25
    json.dumps(out)

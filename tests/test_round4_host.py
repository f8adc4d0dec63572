"""Round-4 host-side checks: the staged reference (oracle/_ref), bench.py's in-run verification and its reference baseline."""
import importlib.util
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import ref_harness, stage_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
  spec = importlib.util.spec_from_file_location('bench', os.path.join(ROOT, 'bench.py'))
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  return mod


@pytest.mark.skipif(not os.path.isdir('/root/reference/spriteworld'), reason='staging needs the reference sources')
def test_staged_reference_is_bytecode_only_and_steps_like_the_source_tree():
  """oracle/stage_ref.py compiles /root/reference into sourceless .pyc files (no source copied); a fresh interpreter that
  only sees oracle/_ref runs the reference Environment and produces the frames the source tree produces."""
  manifest = stage_ref.stage()
  assert manifest and 'spriteworld/environment.py' in manifest['modules']
  for d, _, files in os.walk(stage_ref.OUT):
    assert not [f for f in files if f.endswith('.py')], (d, files)           # never sources
  code = """
import sys, zlib, importlib
import numpy as np
sys.path.insert(0, %r)
from oracle import ref_harness
ref = ref_harness.load_reference()
np.random.seed(3)
env = ref.environment.Environment(**importlib.import_module('spriteworld.configs.cobra.sorting').get_config('train'))
env.reset()
crc = 0
for t in range(12):
  ts = env.step(env.action_space.sample())
  crc = zlib.crc32(ts.observation['image'].tobytes(), crc)
print(ref_harness.reference_kind(), crc)
""" % ROOT
  outs = []
  for root in ('/root/reference', stage_ref.OUT):
    env = dict(os.environ, SPRITEWORLD_REFERENCE=root, PYTHONDONTWRITEBYTECODE='1')
    outs.append(subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, check=True).stdout.split())
  assert outs[0][0] == 'source' and outs[1][0] == 'bytecode'
  assert outs[0][1] == outs[1][1]


def test_bench_in_run_verification_catches_a_wrong_environment():
  """bench.verify_against_oracle on a run of the emulated kernel: 0 mismatches; one corrupted frame / position -> counted."""
  from spriteworld_amd import workloads
  from tests import _emu_engine
  bench = _bench()
  n, warmup, steps = 24, 3, 9
  cfg, pool, sample = workloads.build('cluster_s5', n, episodes_per_env=4, seed=0, anti_aliasing=5)
  eng = _emu_engine.EmuEngine(cfg, pool)
  rng = np.random.default_rng(2000)
  acts = [sample(rng) for _ in range(bench.N_ACTION_SETS)]
  for k in (warmup, steps):
    for i in range(k):
      eng.step(acts[i % bench.N_ACTION_SETS])
  idx = np.sort(np.random.default_rng(99).choice(n, size=8, replace=False))
  got, st = eng.outputs_host(), eng.state()
  smp = dict(idx=idx, cfg=cfg, pool=pool, warmup=warmup, steps=steps, actions=[np.ascontiguousarray(a[idx]) for a in acts],
             got={k: got[k][idx].copy() for k in ('obs', 'reward', 'step_type', 'success', 'discount')},
             state={k: st[k][idx].copy() for k in ('x', 'y', 'step_count', 'episode', 'n_sprites')})
  ok = bench.verify_against_oracle(smp)
  assert ok['verified_envs'] == 8 and ok['mismatches'] == 0 and ok['frame_bytes_differing'] == 0, ok
  json.dumps(ok)
  smp['got']['obs'][2, 5, 7, 1] ^= 1
  smp['state']['x'][5, 0] = np.nextafter(smp['state']['x'][5, 0], 2.0)
  bad = bench.verify_against_oracle(smp)
  assert bad['mismatches'] == 2 and bad['frame_bytes_differing'] == 1 and bad['frame_max_abs_diff'] == 1, bad


@pytest.mark.skipif(not ref_harness.reference_available(), reason='reference not present')
def test_bench_reference_cpu_baseline_block():
  """The cpu_baseline leg that runs the unmodified reference (kind "reference"), on a tiny sample."""
  bench = _bench()
  block, why = bench.reference_cpu_baseline(envs_per_core=1, steps=6, warmup=2, timeout_s=200)
  assert why is None, why
  assert block['kind'] == 'reference' and block['cores'] == bench.usable_cores() and block['value'] > 0
  assert block['reference']['third_party']['pillow'] and block['cpu'] is not None
  json.dumps(block)

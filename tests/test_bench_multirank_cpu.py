"""bench.py's N > 1 path, run unattended on the CPU: `python -m torch.distributed.run --nproc-per-node 2 ... --gpus 2` with the
gloo backend and the kernel source executed on the host in place of the HIP engine (tests/_bench_dry_run.py).  No 8-GPU node
has been available to the build: the first time this control flow meets RCCL must not be the first time it runs at all.
Asserted: ONE JSON line from rank 0 of the contract's shape, world_size / per_rank, value = all ranks' env-steps / the
max-over-ranks time; a rank that fails before the gate and a rank that fails inside the timed region both give ONE line with
`rank_errors` and no hang; --gather-obs runs both schedules of the observation all-gather."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENVS, STEPS, WARMUP = 8, 3, 2


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  return port


def _launch(extra_args=(), fault=None, fault_at=None, timeout=600):
  from tests.emu import build_emu
  build_emu.build()                      # (once, here: the two ranks then only load it)
  env = dict(os.environ, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='1', PYTHONPATH=ROOT)
  env.pop('SWB_DRY_FAULT', None)
  if fault:
    env['SWB_DRY_FAULT'] = fault
    env['SWB_DRY_FAULT_AT'] = str(fault_at or 0)
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
         '--master-port', str(_free_port()), os.path.join(ROOT, 'tests', '_bench_dry_run.py'), '--gpus', '2', '--steps', str(STEPS),
         '--warmup', str(WARMUP), '--envs-per-gpu', str(ENVS), '--no-cpu-baseline'] + list(extra_args)
  proc = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
  lines = [l for l in proc.stdout.splitlines() if l.startswith('{')]
  return proc, lines


def test_two_ranks_print_one_line_of_the_contracts_shape():
  proc, lines = _launch(['--ramp-ms', '1'])
  assert proc.returncode == 0, proc.stderr[-2000:]
  assert len(lines) == 1, proc.stdout[-2000:]
  d = json.loads(lines[0])
  for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline'):
    assert key in d, key
  assert d['n_gpus'] == 2 and d['steps'] == STEPS and d['warmup'] == WARMUP and d['scaling'] == 'weak'
  assert d['world_size'] == 2 and d['backend'] == 'gloo'
  assert len(d['per_rank']) == 2 and sorted(r['rank'] for r in d['per_rank']) == [0, 1]
  assert all(r['error'] is None and r['env_errors'] == 0 for r in d['per_rank'])
  # value = the env-steps of ALL ranks / the max-over-ranks wall time of the timed region
  elapsed = d['ms_per_step'] * STEPS / 1e3
  assert abs(d['value'] - 2 * ENVS * STEPS / elapsed) / d['value'] < 1e-9
  assert elapsed >= max(r['elapsed_s'] for r in d['per_rank']) * (1 - 1e-9)
  # the in-run oracle check of rank 0's sampled environments ran on the emulated kernel: no mismatch
  assert d['verified_envs'] == ENVS and d['mismatches'] == 0
  # the same clock ramp on every rank (round-5 advice): reported with the line, and the cold figure beside it
  assert d['warmup_effective']['timed_engine_warmup_steps'] == WARMUP and d['warmup_effective']['clock_ramp_steps'] >= 16
  assert d['cold']['ms_per_step'] > 0
  assert 'extra' not in d and 'cpu_baseline' not in d          # N > 1: neither
  assert 'np.random.seed(1000 + env)' in d['data']             # SURVEY 8d's input protocol, on every rank (global env indices)


def test_a_rank_that_cannot_set_up_gives_one_line_with_rank_errors_and_no_hang():
  proc, lines = _launch(['--ramp-ms', '0'], fault='setup:1', timeout=300)
  assert proc.returncode == 0, proc.stderr[-2000:]
  assert len(lines) == 1, proc.stdout[-2000:]
  d = json.loads(lines[0])
  assert d['value'] is None and d['world_size'] == 2
  assert [e['rank'] for e in d['rank_errors']] == [1] and 'injected' in d['rank_errors'][0]['error']


def test_a_rank_that_fails_in_the_timed_region_gives_one_line_with_rank_errors_and_no_hang():
  proc, lines = _launch(['--ramp-ms', '0'], fault='timed:1', fault_at=WARMUP + 2, timeout=300)
  assert proc.returncode == 0, proc.stderr[-2000:]
  assert len(lines) == 1, proc.stdout[-2000:]
  d = json.loads(lines[0])
  assert d['value'] is None
  assert [e['rank'] for e in d['rank_errors']] == [1] and 'injected' in d['rank_errors'][0]['error']


def test_a_rank_that_fails_in_its_warm_up_steps_behind_the_gate_gives_one_line_with_rank_errors_and_no_hang():
  # (the gate comes BEFORE the clock ramp and the warm-up steps since round 6 -- it aligns the ranks in time --, so a failure in
  # the warm-up is carried past the barriers like one in the timed region)
  proc, lines = _launch(['--ramp-ms', '0'], fault='timed:1', fault_at=1, timeout=300)
  assert proc.returncode == 0, proc.stderr[-2000:]
  assert len(lines) == 1, proc.stdout[-2000:]
  d = json.loads(lines[0])
  assert d['value'] is None
  assert [e['rank'] for e in d['rank_errors']] == [1] and 'injected' in d['rank_errors'][0]['error']


def test_gather_obs_runs_both_schedules_on_two_ranks():
  proc, lines = _launch(['--ramp-ms', '0', '--gather-obs', '--no-verify'])
  assert proc.returncode == 0, proc.stderr[-2000:]
  assert len(lines) == 1, proc.stdout[-2000:]
  d = json.loads(lines[0])
  g = d['obs_allgather']
  assert set(g) == {'ring', 'direct'}
  for m in ('ring', 'direct'):
    assert g[m]['method'] == m and g[m]['gathered_bytes_per_step'] == 2 * ENVS * 64 * 64 * 3
    assert g[m]['env_steps_per_s'] > 0 and g[m]['step_ms'] > 0 and g[m]['gather_ms'] >= 0

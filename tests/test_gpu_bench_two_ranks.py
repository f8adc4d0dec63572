"""bench.py's N > 1 path with the REAL engine, on the one GPU a test box has: two ranks under `torch.distributed.run`, both on
cuda:0 (SWB_BENCH_ONE_DEVICE=1: gloo instead of RCCL, which refuses two ranks on one device).  The CPU suite runs the same
control flow on the emulated kernel (tests/test_bench_multirank_cpu.py); this is the gate, the clock ramp that lasts until
every rank has arrived, the barriers and the max-over-ranks timing around HIP launches.  No 8-GPU node has been available to
the build: what cannot be covered here is RCCL itself."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  return port


def _launch(extra):
  env = dict(os.environ, MASTER_ADDR='127.0.0.1', SWB_BENCH_ONE_DEVICE='1', OMP_NUM_THREADS='1', PYTHONPATH=ROOT)
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
         '--master-port', str(_free_port()), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '6', '--warmup', '3',
         '--envs-per-gpu', '512', '--ramp-ms', '20'] + extra
  proc = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
  return proc, [l for l in proc.stdout.splitlines() if l.startswith('{')]


def test_two_ranks_on_one_device_print_one_verified_line():
  proc, lines = _launch([])
  assert proc.returncode == 0, proc.stderr[-2000:]
  assert len(lines) == 1, proc.stdout[-2000:]
  d = json.loads(lines[0])
  assert d['n_gpus'] == 2 and d['world_size'] == 2 and d['backend'] == 'gloo' and d['scaling'] == 'weak'
  assert sorted(r['rank'] for r in d['per_rank']) == [0, 1] and all(r['error'] is None and r['env_errors'] == 0 for r in d['per_rank'])
  elapsed = d['ms_per_step'] * 6 / 1e3
  assert abs(d['value'] - 2 * 512 * 6 / elapsed) / d['value'] < 1e-9
  assert elapsed >= max(r['elapsed_s'] for r in d['per_rank']) * (1 - 1e-9)
  assert d['verified_envs'] == 64 and d['mismatches'] == 0 and d['frame_bytes_differing'] == 0      # rank 0's run against the oracle
  assert d['warmup_effective']['clock_ramp_steps'] >= 8 and d['cold']['ms_per_step'] > 0
  assert 'ramp_error' not in json.dumps(d)
  assert d['roofline']['kernel'].startswith('swb_cover_kernel<10>') and 'extra' not in d and 'cpu_baseline' not in d


def test_two_ranks_gather_their_observations_with_both_schedules():
  proc, lines = _launch(['--gather-obs', '--no-verify'])
  assert proc.returncode == 0, proc.stderr[-2000:]
  assert len(lines) == 1, proc.stdout[-2000:]
  g = json.loads(lines[0])['obs_allgather']
  for m in ('ring', 'direct'):
    assert g[m]['method'] == m and g[m]['gathered_bytes_per_step'] == 2 * 512 * 64 * 64 * 3 and g[m]['step_ms'] > 0

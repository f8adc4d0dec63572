"""Runs bench.py's main() WITHOUT a GPU (TEST INFRASTRUCTURE ONLY): the control flow of the N > 1 path -- rendezvous, the
gate, barriers, max-over-ranks timing, per-rank records, the failure paths, --gather-obs -- under `torch.distributed.run`
with the gloo backend, before it ever meets RCCL on an 8-GPU node.

What is substituted, in THIS process only (bench.py itself has no dry-run mode and is not edited by this file):
  * spriteworld_amd.engine.Engine -> the kernel SOURCE executed on the host (tests/_emu_engine.EmuTorchEngine) with the timing
    calls of the real engine answered by the host clock;
  * the handful of torch.cuda entry points bench.py touches (synchronize, Event, set_device, get_device_name) -> host no-ops /
    host clock.
The figures it prints mean nothing; the tests (tests/test_bench_multirank_cpu.py) assert the SHAPE of the line.

Fault injection (environment):  SWB_DRY_FAULT=setup:R   rank R fails while it builds its engine (before the gate)
                                SWB_DRY_FAULT=timed:R   rank R's engine raises inside the timed region
usage: python -m torch.distributed.run ... tests/_bench_dry_run.py --gpus 2 --steps K --warmup W [bench.py's flags]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

import torch  # noqa: E402

from spriteworld_amd import engine as _engine  # noqa: E402
from tests import _emu_engine  # noqa: E402

RANK = int(os.environ.get('RANK', '0'))
FAULT = os.environ.get('SWB_DRY_FAULT', '')
_built = [0]


class DryEngine(_emu_engine.EmuTorchEngine):
  """EmuTorchEngine + the timing surface of engine.Engine (host clock) + the injected faults."""

  def __init__(self, cfg, pool, device=0):
    _built[0] += 1
    self._first = _built[0] == 1                 # the timed engine of this rank is the first one bench.py builds
    if FAULT == 'setup:%d' % RANK and self._first:
      raise RuntimeError('injected: rank %d cannot build its engine' % RANK)
    _emu_engine.EmuTorchEngine.__init__(self, cfg, pool, device)
    self._timing, self._ms, self._k, self._calls = False, 0.0, 0, 0

  def step(self, actions, render=True):
    self._calls += 1
    if FAULT == 'timed:%d' % RANK and self._first and self._calls == int(os.environ['SWB_DRY_FAULT_AT']):
      raise RuntimeError('injected: rank %d fails in the timed region' % RANK)
    t0 = time.perf_counter()
    _emu_engine.EmuTorchEngine.step(self, actions, render=render)
    if self._timing:
      self._ms += (time.perf_counter() - t0) * 1e3
      self._k += 1

  def timing(self, enable):
    self._timing = bool(enable)

  def step_time_ms(self):
    return self._ms, self._k

  def kernel_times_ms(self):
    return 0.4 * self._ms, 0.6 * self._ms, self._k


class _Event(object):

  def __init__(self, enable_timing=False):
    self.t = None

  def record(self, stream=None):
    self.t = time.perf_counter()

  def elapsed_time(self, other):
    return (other.t - self.t) * 1e3


def main():
  _engine.Engine = DryEngine
  torch.cuda.synchronize = lambda *a, **k: None
  torch.cuda.set_device = lambda *a, **k: None
  torch.cuda.get_device_name = lambda *a, **k: 'dry run (host emulation of the kernel source)'
  torch.cuda.Event = _Event
  os.environ['SWB_BENCH_ONE_DEVICE'] = '1'        # bench.py: gloo, every rank on "device 0"
  import importlib.util
  spec = importlib.util.spec_from_file_location('bench', os.path.join(ROOT, 'bench.py'))
  bench = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(bench)
  bench.main()


if __name__ == '__main__':
  main()

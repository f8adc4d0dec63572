#!/usr/bin/env python
"""Exact dynamic event counts of the step kernel per environment and step, from the host emulation (tests/emu) built with
event counters (SWB_EMU_STATS=1): canvas rows that reach the resampling loop, row runs after folding, spans per run,
sprite x batch passes of the coverage phase ... -- the inputs of the cost model in DESIGN.md section 3.
usage: python tools/emu_stats.py [WORKLOAD] [N_ENVS] [STEPS] [AA]"""
import ctypes
import os
import sys

os.environ['SWB_EMU_STATS'] = '1'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from spriteworld_amd import workloads  # noqa: E402
from tests import _emu_engine  # noqa: E402
from tests.emu import build_emu  # noqa: E402


def main():
  name = sys.argv[1] if len(sys.argv) > 1 else 'cluster_s5'
  n_envs = int(sys.argv[2]) if len(sys.argv) > 2 else 64
  steps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
  aa = int(sys.argv[4]) if len(sys.argv) > 4 else 5
  cfg, pool, sample = workloads.build(name, n_envs, episodes_per_env=4, seed=0, anti_aliasing=aa)
  eng = _emu_engine.EmuEngine(cfg, pool)
  lib = eng.lib
  lib.emu_stats.restype = ctypes.c_long
  rng = np.random.default_rng(2000)
  eng.step(sample(rng))                       # the reset step
  lib.emu_stats(0, 1)
  for _ in range(steps):
    eng.step(sample(rng))
  total = n_envs * steps
  print('%s, anti_aliasing %d, %d environments x %d steps: events per environment and step' % (name, aa, n_envs, steps))
  vals = {c: lib.emu_stats(i, 0) / total for i, c in enumerate(build_emu._COUNTERS)}
  for c in build_emu._COUNTERS:
    print('  %-22s %9.2f' % (c, vals[c]))
  if vals['p3_row_runs']:
    print('  rows per run %.2f, spans per run %.2f, runs per non-empty row %.2f' % (
        vals['p3_rows_in_runs'] / vals['p3_row_runs'], vals['p3_spans'] / vals['p3_row_runs'],
        vals['p3_row_runs'] / max(vals['p3_nonempty_rows'], 1e-9)))


if __name__ == '__main__':
  main()

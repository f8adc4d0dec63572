#!/usr/bin/env python
"""Exact dynamic event counts of the step kernel per environment and step, from the host emulation (tests/emu) built with
event counters (SWB_EMU_STATS=1): canvas rows that reach the resampling loop, row runs after folding, spans per run,
sprite x batch passes of the coverage phase ... -- the inputs of the cost model in DESIGN.md section 3.
usage: python tools/emu_stats.py [WORKLOAD] [N_ENVS] [STEPS] [AA] [BANDS]   (BANDS: bands of output rows in the
resample kernel, default 1 -- what the engine uses from 8192 environments)"""
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
  os.environ['SWB_BANDS'] = sys.argv[5] if len(sys.argv) > 5 else '1'
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
    print('  rows per run %.2f, spans per run %.2f, 8-byte units per run %.2f' % (
        vals['p3_rows_in_runs'] / vals['p3_row_runs'], vals['p3_spans'] / vals['p3_row_runs'], vals['run_units'] / vals['p3_row_runs']))
  return vals


def resample_valu_model(vals):
  """The resample kernel's vector instructions per environment if nothing but the arithmetic of its algorithm were
  issued (swb_kernels.hip.inc, resample loop): per run 2 med3 + 2 address + 1 readlane + 1 sub + 3 mad for the first
  span, 3 shifts + 3 med3 (clip), 18 mads (six output rows in flight x three channels) = 33; 10 per further span; a
  finished row that received something 3 reads + 3 restarts + 6 clip + 2 pack + 1 DPP + 1 perm + 1 offset = 17, an
  untouched one 1; ~150 per wave of set-up.  (Round 6: the kernel issues 9 per further span and 16 per finished row -- the model
  is kept as it was, so that the figures of the rounds compare; a scene of many spans per run measures slightly below it.)"""
  runs, spans = vals['p3_row_runs'], vals['p3_spans']
  done, clean = vals['p3_completed_rows'], vals['p3_clean_rows']
  return 33 * runs + 10 * (spans - runs) + 17 * (done - clean) + 1 * clean + 150


if __name__ == '__main__':
  v = main()
  print('  resample kernel, cost-model minimum: %.0f vector instructions per environment' % resample_valu_model(v))

#!/usr/bin/env python
"""Kernel times of a step (HIP events on the launch stream: whole step, cover kernel, resample / fill kernel) for a
list of workloads.  usage: python tools/quick_bench.py WORKLOAD:N_ENVS:AA[:BANDS] ...   (BANDS -> SWB_BANDS)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from spriteworld_amd import engine, workloads  # noqa: E402


def run(name, n, aa, bands, steps=40, warmup=10):
  if bands:
    os.environ['SWB_BANDS'] = str(bands)
  else:
    os.environ.pop('SWB_BANDS', None)
  cfg, pool, sample = workloads.build(name, n, episodes_per_env=4, seed=0, anti_aliasing=aa)
  eng = engine.Engine(cfg, pool, device=0)
  rng = np.random.default_rng(2000)
  acts = [torch.as_tensor(sample(rng), device=eng.device) for _ in range(16)]
  for i in range(warmup):
    eng.step(acts[i % 16])
  torch.cuda.synchronize()
  eng.timing(True)
  for i in range(steps):
    eng.step(acts[i % 16])
  torch.cuda.synchronize()
  tot, k = eng.step_time_ms()
  a, b, _ = eng.kernel_times_ms()
  v = eng.variant()
  err = int(eng.error.max().item())
  eng.close()
  print('%-14s N=%-6d AA=%d bands=%d  step %.4f ms  cover %.4f  %s %.4f   (%.1f M env-steps/s, errors %d)' %
        (name, n, aa, v['n_bands'], tot / k, a / k, v['kernel'], b / k, n / (tot / k) / 1e3, err), flush=True)


if __name__ == '__main__':
  for spec in sys.argv[1:]:
    f = spec.split(':')
    run(f[0], int(f[1]), int(f[2]), int(f[3]) if len(f) > 3 else 0)

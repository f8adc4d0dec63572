#!/usr/bin/env python
"""Experiment: what the three HIP events per step of the engine's timing mode cost a run of back-to-back steps (wall time per step
with swb_timing_enable on / off, interleaved).  usage: python tools/exp_timing_overhead.py [ENVS] [STEPS]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from spriteworld_amd import engine, workloads  # noqa: E402


def main():
  n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
  steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
  cfg, pool, sample = workloads.build('cluster_s5', n, episodes_per_env=4, seed=0, anti_aliasing=5)
  eng = engine.Engine(cfg, pool, device=0)
  rng = np.random.default_rng(2000)
  acts = [torch.as_tensor(sample(rng), device=eng.device) for _ in range(16)]
  for i in range(50):
    eng.step(acts[i % 16])
  torch.cuda.synchronize()
  for rep in range(4):
    for timing in (False, True):
      eng.timing(timing)
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      for i in range(steps):
        eng.step(acts[i % 16])
      torch.cuda.synchronize()
      dt = time.perf_counter() - t0
      extra = ''
      if timing:
        tot, k = eng.step_time_ms()
        extra = '  (events: %.4f ms per step)' % (tot / k)
      print('timing %-5s  wall %.4f ms per step  %.2f M env-steps/s%s' % (timing, dt / steps * 1e3, n * steps / dt / 1e6, extra), flush=True)
  eng.close()


if __name__ == '__main__':
  main()

#!/usr/bin/env python
"""Per-launch kernel times of the first launches of a FRESH engine (HIP events on the launch stream): how many launches do
the adaptive parts of the dispatch (bucket widths, priority levels, cover order) and the device itself (clocks, caches) need
before a step costs what the hundredth costs?   usage: python tools/launch_convergence.py [workload] [envs] [aa] [launches]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def main():
  import torch
  from spriteworld_amd import engine, workloads
  name = sys.argv[1] if len(sys.argv) > 1 else 'cluster_s5'
  n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
  aa = int(sys.argv[3]) if len(sys.argv) > 3 else 5
  launches = int(sys.argv[4]) if len(sys.argv) > 4 else 48
  out = {'workload': name, 'envs': n, 'anti_aliasing': aa, 'engines': []}
  for trial in range(2):                       # the second engine starts on a device that is already warm
    cfg, pool, sample = workloads.build(name, n, episodes_per_env=4, seed=trial, anti_aliasing=aa)
    eng = engine.Engine(cfg, pool)
    rng = np.random.default_rng(2000 + trial)
    acts = [torch.as_tensor(sample(rng), device=eng.device) for _ in range(16)]
    eng.timing(True)
    step_ms, cover_ms = [], []
    prev, prev_c = 0.0, 0.0
    for i in range(launches):
      eng.step(acts[i % 16])
      torch.cuda.synchronize()
      total, _ = eng.step_time_ms()
      c, _, _ = eng.kernel_times_ms()
      step_ms.append(round(total - prev, 5))
      cover_ms.append(round(c - prev_c, 5))
      prev, prev_c = total, c
    eng.close()
    tail = float(np.mean(step_ms[-16:]))
    out['engines'].append({'trial': trial, 'step_ms': step_ms, 'cover_ms': cover_ms, 'mean_last_16': tail,
                           'first_launch_within_1pct': next((i for i in range(launches) if max(step_ms[i:i + 4]) <= 1.01 * tail), None)})
  # control: a fixed, compute-bound library kernel (fp32 matmul 2048^3) timed the same way after one second of idling -- if its
  # launches speed up over the same tens of milliseconds, what converges is the device (clocks), not the engine's dispatch
  import time
  a = torch.randn(2048, 2048, device='cuda')
  b = torch.randn(2048, 2048, device='cuda')
  torch.mm(a, b)
  torch.cuda.synchronize()
  control = []
  for trial in range(2):
    time.sleep(1.0)
    ms = []
    for i in range(launches):
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      torch.mm(a, b)
      e1.record()
      torch.cuda.synchronize()
      ms.append(round(e0.elapsed_time(e1), 5))
    control.append({'trial': trial, 'matmul_ms': ms, 'mean_first_8': float(np.mean(ms[:8])), 'mean_last_16': float(np.mean(ms[-16:]))})
  out['control_fp32_matmul_2048'] = control
  print(json.dumps(out))


if __name__ == '__main__':
  main()

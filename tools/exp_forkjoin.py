#!/usr/bin/env python
"""Experiment: the headline batch as G sub-batches stepped on G internal HIP streams, with the fork / join that keeps one
caller stream's ordering (an event recorded on the caller's stream that every internal stream waits for, and an event per internal
stream that the caller's stream waits for), against one launch per step and against unordered stream groups.
usage: python tools/exp_forkjoin.py [ENVS] [STEPS]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from spriteworld_amd import engine, workloads  # noqa: E402


def build(n, groups):
  engs, acts = [], []
  for g in range(groups):
    cfg, pool, sample = workloads.build('cluster_s5', n // groups, episodes_per_env=4, seed=g, anti_aliasing=5)
    engs.append(engine.Engine(cfg, pool))
    rng = np.random.default_rng(2000 + g)
    acts.append([torch.as_tensor(sample(rng), device=engs[g].device) for _ in range(16)])
  return engs, acts


def run(n, groups, mode, steps, warmup=20):
  engs, acts = build(n, groups)
  main = torch.cuda.current_stream()
  streams = [torch.cuda.Stream() for _ in range(groups)]
  fork = torch.cuda.Event()
  joins = [torch.cuda.Event() for _ in range(groups)]

  def step(i):
    if mode == 'forkjoin':
      fork.record(main)
      for g in range(groups):
        streams[g].wait_event(fork)
        with torch.cuda.stream(streams[g]):
          engs[g].step(acts[g][i % 16])
        joins[g].record(streams[g])
      for g in range(groups):
        main.wait_event(joins[g])
    elif mode == 'unordered':
      for g in range(groups):
        with torch.cuda.stream(streams[g]):
          engs[g].step(acts[g][i % 16])
    else:
      for g in range(groups):
        engs[g].step(acts[g][i % 16])
  for i in range(warmup):
    step(i)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for i in range(steps):
    step(i)
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  for e in engs:
    e.close()
  print('%-10s groups %d: %.4f ms per step, %.2f M env-steps/s' % (mode, groups, dt / steps * 1e3, n * steps / dt / 1e6), flush=True)


if __name__ == '__main__':
  n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
  steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
  for rep in range(2):
    run(n, 1, 'serial', steps)
    run(n, 2, 'serial', steps)
    run(n, 2, 'unordered', steps)
    run(n, 2, 'forkjoin', steps)
    run(n, 4, 'forkjoin', steps)
    run(n, 4, 'unordered', steps)

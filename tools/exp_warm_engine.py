#!/usr/bin/env python
"""Why is the 20-step bench line 4 % under the 200-step one?  (round-5 review, weak point 7)

A fresh engine's step time falls over its first ~40 launches (tools/launch_convergence.py), and so does a compute-bound control
kernel after a second of idling.  Is that the DEVICE coming out of idle, or the ENGINE (dispatch state, first touches of its
buffers)?  Three engines of the headline workload are built first; then, without a host-side gap between them,
  A   steps 120 launches out of idle          (device cold, engine fresh)
  B   steps 120 launches right behind A        (device hot,  engine fresh)
  A   steps 120 more launches right behind B   (device hot,  engine warm)
  C   steps 120 launches after 1 s of idling   (device cold again, engine fresh)
each timed in blocks of 5 launches with HIP events on the stream.  usage: python tools/exp_warm_engine.py [OUT.json]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from spriteworld_amd import engine, workloads  # noqa: E402

BLOCK, LAUNCHES = 5, 120


def build(seed):
  cfg, pool, sample = workloads.build('cluster_s5', 8192, episodes_per_env=4, seed=seed, anti_aliasing=5)
  eng = engine.Engine(cfg, pool, device=0)
  rng = np.random.default_rng(2000 + seed)
  acts = [torch.as_tensor(sample(rng), device=eng.device) for _ in range(16)]
  return eng, acts


def run(eng, acts, first):
  evs = [torch.cuda.Event(enable_timing=True) for _ in range(LAUNCHES // BLOCK + 1)]
  evs[0].record()
  for i in range(LAUNCHES):
    eng.step(acts[(first + i) % 16])
    if (i + 1) % BLOCK == 0:
      evs[(i + 1) // BLOCK].record()
  return evs


def main():
  engs = [build(s) for s in range(3)]
  torch.cuda.synchronize()
  time.sleep(1.0)
  ev_a = run(*engs[0], 0)
  ev_b = run(*engs[1], 0)
  ev_a2 = run(*engs[0], LAUNCHES)
  torch.cuda.synchronize()
  time.sleep(1.0)
  ev_c = run(*engs[2], 0)
  torch.cuda.synchronize()
  out = {}
  for name, evs in (('A: device cold, engine fresh', ev_a), ('B: device hot, engine fresh', ev_b), ('A again: device hot, engine warm', ev_a2),
                    ('C: after 1 s idle, engine fresh', ev_c)):
    ms = [evs[k].elapsed_time(evs[k + 1]) / BLOCK for k in range(len(evs) - 1)]
    out[name] = ms
    print('%-34s launches 0-4 %.4f  5-24 %.4f  25-59 %.4f  60-119 %.4f ms/step' % (
        name, ms[0], float(np.mean(ms[1:5])), float(np.mean(ms[5:12])), float(np.mean(ms[12:]))), flush=True)
  for e, _ in engs:
    e.close()
  if len(sys.argv) > 1:
    json.dump({'block': BLOCK, 'ms_per_step_by_block': out}, open(sys.argv[1], 'w'), indent=1)


if __name__ == '__main__':
  main()

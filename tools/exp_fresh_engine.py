#!/usr/bin/env python
"""Why are launches 5 .. 24 of a FRESH engine 1.6 % slower than its later ones even on a hot device (tools/exp_warm_engine.py)?
Candidates: the trim of the run lists after the third rendering launch (new allocations), the adaptive parts of the dispatch.
Four engines of the headline workload are built first; a fifth keeps the device hot for 0.5 s; then, back to back, 60 launches
each in blocks of 5 (HIP events): (a) as shipped, (b) trim switched off, (c) trimmed before the first launch of the block
sequence (after 3 priming launches + 40 more to settle), (d) SWB_NO_COVER_ORDER / SWB_NO_PRIO.
usage: python tools/exp_fresh_engine.py [OUT.json]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from spriteworld_amd import engine, workloads  # noqa: E402

BLOCK, LAUNCHES = 5, 60


def build(seed, trim_after=3, env=None):
  old = {}
  for k, v in (env or {}).items():
    old[k] = os.environ.get(k)
    os.environ[k] = v
  cfg, pool, sample = workloads.build('cluster_s5', 8192, episodes_per_env=4, seed=seed, anti_aliasing=5)
  eng = engine.Engine(cfg, pool, device=0)
  eng.TRIM_AFTER = trim_after
  for k, v in old.items():
    if v is None:
      os.environ.pop(k, None)
    else:
      os.environ[k] = v
  rng = np.random.default_rng(2000 + seed)
  acts = [torch.as_tensor(sample(rng), device=eng.device) for _ in range(16)]
  return eng, acts


def run(eng, acts, n=LAUNCHES):
  evs = [torch.cuda.Event(enable_timing=True) for _ in range(n // BLOCK + 1)]
  evs[0].record()
  for i in range(n):
    eng.step(acts[i % 16])
    if (i + 1) % BLOCK == 0:
      evs[(i + 1) // BLOCK].record()
  return evs


def main():
  out = {}
  for rep in range(2):
    variants = [('as shipped (trim after launch 3)', build(1)), ('no trim', build(2, trim_after=10 ** 9)),
                ('no cover order, no priorities', build(3, env={'SWB_NO_COVER_ORDER': '1', 'SWB_NO_PRIO': '1', 'SWB_NO_COVER_PRIO': '1'})),
                ('trim after launch 1', build(4, trim_after=1))]
    hot, hot_acts = build(9)
    torch.cuda.synchronize()
    for i in range(2800):                           # ~0.5 s of load
      hot.step(hot_acts[i % 16])
    evs = [(name, run(e, a)) for name, (e, a) in variants]
    torch.cuda.synchronize()
    for name, ev in evs:
      ms = [ev[k].elapsed_time(ev[k + 1]) / BLOCK for k in range(len(ev) - 1)]
      out.setdefault(name, []).append([round(x, 4) for x in ms])
      print('%-34s %s' % (name, ' '.join('%.4f' % x for x in ms)), flush=True)
    for _, (e, _) in variants:
      e.close()
    hot.close()
  if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], 'w'), indent=1)


if __name__ == '__main__':
  main()

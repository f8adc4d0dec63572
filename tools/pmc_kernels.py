#!/usr/bin/env python
"""Per-kernel PMC averages of a short run (one rocprofv3 --pmc pass per invocation; counters in their own process).

  rocprofv3 --pmc C1 C2 ... -d DIR -o p -- python tools/pmc_kernels.py run WORKLOAD N_ENVS AA [STEPS]
  python tools/pmc_kernels.py report DB [DB ...]        # JSON: {kernel: {counter: mean per dispatch, 'dispatches': n}}
Dispatches of the first 4 steps (the reset step, the first episodes' start) are left out of the averages."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(name, n, aa, steps):
  import numpy as np
  import torch
  from spriteworld_amd import engine, workloads
  cfg, pool, sample = workloads.build(name, n, episodes_per_env=4, seed=0, anti_aliasing=aa)
  eng = engine.Engine(cfg, pool, device=0)
  rng = np.random.default_rng(2000)
  acts = [torch.as_tensor(sample(rng), device=eng.device) for _ in range(8)]
  for i in range(steps):
    eng.step(acts[i % 8])
  torch.cuda.synchronize()
  eng.close()


def report(dbs):
  import sqlite3
  import numpy as np
  out = {}
  for db in dbs:
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select dispatch_id, kernel_name, counter_name, value from counters_collection order by dispatch_id"))
    per = {}
    for did, kname, cname, val in rows:
      if 'swb_' not in kname:
        continue
      k = kname.split('(')[0].replace('void ', '')
      per.setdefault(k, {}).setdefault(did, {}).setdefault(cname, 0.0)
      per[k][did][cname] += val
    for k, d in per.items():
      ids = sorted(d)[4:]
      o = out.setdefault(k, {})
      o['dispatches'] = len(ids)
      for c in sorted(set(c for i in ids for c in d[i])):
        o[c] = float(np.mean([d[i].get(c, 0.0) for i in ids]))
  print(json.dumps(out, indent=1))


if __name__ == '__main__':
  if sys.argv[1] == 'run':
    run(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]) if len(sys.argv) > 5 else 14)
  else:
    report(sys.argv[2:])

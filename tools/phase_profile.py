#!/usr/bin/env python
"""Phase split of the fused step kernel: kernel time and (under rocprofv3 --pmc) instruction counts
with the kernel cut short after each phase (SWB_DEBUG_PHASE, read by swb_create).

  python tools/phase_profile.py time OUT.md            # HIP-event kernel time per (workload, phase)
  rocprofv3 --pmc ... -d DIR -o p -- python tools/phase_profile.py pmc-run ORDER.json
  python tools/phase_profile.py pmc-report ORDER.json DB [DB ...] >> OUT.md

Phases (cumulative, in execution order): 3 = loads + centred paths (P0 loads, P1a); 4 = + action
hit-test / move (P0); 5 = + task reward, termination, outputs (P0); 1 = + canvas edges (P1b);
2 = + coverage of every batch (P2, no resample); 0 = the whole kernel (+ P3 resample, P4 store).
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

PHASES = (3, 4, 5, 1, 2, 0)
PHASE_NAMES = {3: 'loads + centred paths', 4: '+ action/hit-test/move', 5: '+ task/termination', 1: '+ canvas edges (P1b)',
               2: '+ coverage (P2)', 0: '+ resample/store (P3/P4) = all'}
WORKLOADS = (('cluster_s5', 5), ('cluster_s5', 1), ('goal_s5', 5), ('embodied_s12', 5))
N_ENVS = 8192
PMC_LAUNCHES = 12
# PHASE_WORKLOADS="cluster_s5:5,goal_s5:5" PHASE_LIST="2,0" restrict a run (same values for pmc-run and pmc-report)
if os.environ.get('PHASE_WORKLOADS'):
  WORKLOADS = tuple((w.split(':')[0], int(w.split(':')[1])) for w in os.environ['PHASE_WORKLOADS'].split(','))
if os.environ.get('PHASE_LIST'):
  PHASES = tuple(int(v) for v in os.environ['PHASE_LIST'].split(','))


def run(name, aa, phase, steps, warmup, timing):
  import torch
  from spriteworld_amd import engine, workloads
  os.environ['SWB_DEBUG_PHASE'] = str(phase)
  cfg, pool, sample = workloads.build(name, N_ENVS, episodes_per_env=4, seed=0, anti_aliasing=aa)
  eng = engine.Engine(cfg, pool, device=0)
  rng = np.random.default_rng(2000)
  acts = [torch.as_tensor(sample(rng), device=eng.device) for _ in range(8)]
  for i in range(warmup):
    eng.step(acts[i % 8])
  torch.cuda.synchronize()
  if timing:
    eng.timing(True)
  for i in range(steps):
    eng.step(acts[i % 8])
  torch.cuda.synchronize()
  ms = None
  if timing:
    tot, n = eng.step_time_ms()
    ms = tot / max(n, 1)
  eng.close()
  return ms


def main():
  mode = sys.argv[1]
  if mode == 'time':
    lines = ['## Kernel time by phase (HIP events, %d envs, ms per launch; cumulative, then the increment)' % N_ENVS, '',
             '| workload | ' + ' | '.join('%d: %s' % (p, PHASE_NAMES[p]) for p in PHASES) + ' |',
             '|---|' + '---|' * len(PHASES)]
    for name, aa in WORKLOADS:
      ts = [run(name, aa, p, 40, 5, True) for p in PHASES]
      cells = ['%.4f (+%.4f)' % (t, t - (ts[i - 1] if i else 0.0)) for i, t in enumerate(ts)]
      lines.append('| %s AA=%d | ' % (name, aa) + ' | '.join(cells) + ' |')
      print(lines[-1], flush=True)
    open(sys.argv[2], 'a').write('\n'.join(lines) + '\n\n')
  elif mode == 'pmc-run':
    order = []
    for name, aa in WORKLOADS:
      for p in PHASES:
        run(name, aa, p, PMC_LAUNCHES - 2, 2, False)
        order.append([name, aa, p, PMC_LAUNCHES])
    json.dump(order, open(sys.argv[2], 'w'))
  elif mode == 'pmc-json':
    # per-dispatch averages of every counter, per (workload, phase) group, as JSON on stdout
    import sqlite3
    order = json.load(open(sys.argv[2]))
    out = {}
    for db in sys.argv[3:]:
      cur = sqlite3.connect(db).cursor()
      rows = list(cur.execute("select dispatch_id, counter_name, value from counters_collection "
                              "where kernel_name like '%swb_step%' order by dispatch_id"))
      ids = sorted(set(r[0] for r in rows))
      if len(ids) != sum(o[3] for o in order):
        print('dispatch count mismatch in %s: %d vs %d' % (db, len(ids), sum(o[3] for o in order)), file=sys.stderr)
        continue
      group, k = {}, 0
      for gi, o in enumerate(order):
        for j in range(o[3]):
          if j >= 4:
            group[ids[k]] = gi
          k += 1
      acc = {}
      for did, cname, val in rows:
        if did in group:
          acc.setdefault((group[did], cname), []).append(val)
      for (gi, cname), vals in acc.items():
        o = order[gi]
        out.setdefault('%s:%d:%d' % (o[0], o[1], o[2]), {})[cname] = float(np.mean(vals))
    print(json.dumps(out, indent=1))
  elif mode == 'pmc-report':
    import sqlite3
    order = json.load(open(sys.argv[2]))
    per = {}
    for db in sys.argv[3:]:
      cur = sqlite3.connect(db).cursor()
      rows = list(cur.execute("select dispatch_id, counter_name, value from counters_collection "
                              "where kernel_name like '%swb_step%' order by dispatch_id"))
      ids = sorted(set(r[0] for r in rows))
      assert len(ids) == sum(o[3] for o in order), (len(ids), sum(o[3] for o in order))
      group = {}
      k = 0
      for gi, o in enumerate(order):
        for j in range(o[3]):
          if j >= 4:          # skip the reset step and the first episodes' start
            group[ids[k]] = gi
          k += 1
      for did, cname, val in rows:
        if did in group:
          per.setdefault((group[did], cname), []).append(val)
    counters = sorted(set(c for _, c in per))
    print('## Instruction counters by phase (rocprofv3 --pmc, per wave = per environment; cumulative)\n')
    print('| workload | phase | ' + ' | '.join(counters) + ' |')
    print('|---|---|' + '---|' * len(counters))
    for gi, o in enumerate(order):
      vals = ['%.0f' % (np.mean(per[(gi, c)]) / N_ENVS) if (gi, c) in per else '' for c in counters]
      print('| %s AA=%d | %d: %s | ' % (o[0], o[1], o[2], PHASE_NAMES[o[2]]) + ' | '.join(vals) + ' |')
    print()


if __name__ == '__main__':
  main()

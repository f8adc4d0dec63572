#!/usr/bin/env python
"""Can the tail of a launch be filled from the host side?  (round-5 review, item 4)

The round-5 two-rank / one-device run of bench.py gave 50.0 M env-steps/s from two processes of 8192 environments sharing
one MI355X against 44.5 M for one process: this tool separates "two queues fill each other's tails" from "twice the work
per unit of fixed cost".  Interleaved on ONE box, ROUNDS times, all on the headline scene (cluster_s5, anti_aliasing 5):

  one_8192        one process, one engine of 8192 environments, one stream                      (the bench line)
  one_16384       ... of 16384 environments                                                      (the fair partner of two_8192)
  two_4096        two PROCESSES of 4096 environments each, started together                      (same total work as one_8192)
  two_8192        two processes of 8192 each                                                     (the round-5 observation)
  streams_4096    one process, two engines of 4096 on two streams created before any launch      (EnvironmentGroups)
  prio_4096       ... the two streams at different priorities (high / low)
  streams_8192    one process, two engines of 8192 on two streams                                (partner of two_8192)

Every figure is total env-steps / wall time from the first process's start to the last one's end of the timed steps.
usage: python tools/exp_queue_concurrency.py [ROUNDS] [STEPS]      (child mode: ... --child N_ENVS STEPS GO_FILE OUT_FILE)"""
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

WORKLOAD, AA, WARMUP = 'cluster_s5', 5, 30


def build_engine(n, seed):
  import numpy as np
  import torch
  from spriteworld_amd import engine, workloads
  cfg, pool, sample = workloads.build(WORKLOAD, n, episodes_per_env=4, seed=seed, anti_aliasing=AA)
  eng = engine.Engine(cfg, pool, device=0)
  rng = np.random.default_rng(2000 + seed)
  acts = [torch.as_tensor(sample(rng), device=eng.device) for _ in range(16)]
  return eng, acts


def child(n, steps, go_file, out_file):
  import torch
  eng, acts = build_engine(n, int(os.environ.get('SWB_EXP_SEED', '0')))
  for i in range(WARMUP):
    eng.step(acts[i % 16])
  torch.cuda.synchronize()
  with open(out_file + '.ready', 'w') as f:
    f.write('ready')
  while not os.path.exists(go_file):
    pass
  # keep the device busy until the common start time, then time `steps` steps
  with open(go_file) as f:
    txt = f.read()
  while not txt:
    with open(go_file) as f:
      txt = f.read()
  t_start = float(txt)
  i = 0
  while time.time() < t_start:
    eng.step(acts[i % 16]); i += 1
    if i % 8 == 0:
      torch.cuda.synchronize()
  torch.cuda.synchronize()
  t0 = time.time()
  for k in range(steps):
    eng.step(acts[k % 16])
  torch.cuda.synchronize()
  t1 = time.time()
  errs = int(eng.error.max().item())
  with open(out_file, 'w') as f:
    json.dump({'n': n, 'steps': steps, 't0': t0, 't1': t1, 'errors': errs}, f)
  eng.close()


def run_processes(ns, steps):
  d = tempfile.mkdtemp(prefix='swbq')
  go = os.path.join(d, 'go')
  outs, procs = [], []
  for k, n in enumerate(ns):
    out = os.path.join(d, 'out%d.json' % k)
    outs.append(out)
    env = dict(os.environ, SWB_EXP_SEED=str(k))
    procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), '--child', str(n), str(steps), go, out], env=env))
  while not all(os.path.exists(o + '.ready') for o in outs):
    if any(p.poll() not in (None, 0) for p in procs):
      raise RuntimeError('child failed')
    time.sleep(0.01)
  with open(go + '.tmp', 'w') as f:
    f.write(repr(time.time() + 0.25))
  os.rename(go + '.tmp', go)
  for p in procs:
    if p.wait() != 0:
      raise RuntimeError('child failed')
  recs = [json.load(open(o)) for o in outs]
  wall = max(r['t1'] for r in recs) - min(r['t0'] for r in recs)
  overlap = min(r['t1'] for r in recs) - max(r['t0'] for r in recs)
  return {'env_steps_per_s': sum(r['n'] * r['steps'] for r in recs) / wall, 'wall_s': wall,
          'overlap_frac': overlap / wall if len(recs) > 1 else 1.0, 'errors': max(r['errors'] for r in recs)}


def run_streams(n_each, steps, priorities):
  import torch
  streams = [torch.cuda.Stream(device=0, priority=p) for p in priorities]       # created before any launch of these engines
  engs = [build_engine(n_each, g) for g in range(len(streams))]

  def go(k):
    for i in range(k):
      for (eng, acts), st in zip(engs, streams):
        with torch.cuda.stream(st):
          eng.step(acts[i % 16])
  go(WARMUP)
  torch.cuda.synchronize()
  t0 = time.time()
  go(steps)
  torch.cuda.synchronize()
  wall = time.time() - t0
  errs = max(int(e.error.max().item()) for e, _ in engs)
  for e, _ in engs:
    e.close()
  return {'env_steps_per_s': n_each * len(streams) * steps / wall, 'wall_s': wall, 'errors': errs}


def main():
  rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
  steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
  import torch  # noqa: F401  (the parent holds a context too: it runs the stream variants itself)
  cases = [
      ('one_8192', lambda: run_processes([8192], steps)),
      ('one_16384', lambda: run_processes([16384], steps)),
      ('two_4096', lambda: run_processes([4096, 4096], steps)),
      ('two_8192', lambda: run_processes([8192, 8192], steps)),
      ('streams_4096', lambda: run_streams(4096, steps, (0, 0))),
      ('prio_4096', lambda: run_streams(4096, steps, (-1, 0))),
      ('streams_8192', lambda: run_streams(8192, steps, (0, 0))),
  ]
  table = {name: [] for name, _ in cases}
  for r in range(rounds):
    for name, fn in cases:
      rec = fn()
      table[name].append(rec)
      print('round %d %-13s %7.2f M env-steps/s  wall %.4f s  %s errors %d' % (
          r, name, rec['env_steps_per_s'] / 1e6, rec['wall_s'],
          ('overlap %.2f' % rec['overlap_frac']) if 'overlap_frac' in rec else '', rec['errors']), flush=True)
  print(json.dumps({'steps': steps, 'rounds': rounds, 'workload': WORKLOAD, 'anti_aliasing': AA, 'results': table}))


if __name__ == '__main__':
  if len(sys.argv) > 1 and sys.argv[1] == '--child':
    child(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5])
  else:
    main()

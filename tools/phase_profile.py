#!/usr/bin/env python
"""Where a step's time goes: the two kernels (HIP events on the launch stream) and, inside the cover kernel, its phases --
from builds of it cut short after a phase (tools/overlays/phase_cut.py; one process per build, the library is chosen
through SWB_LIBRARY).

  for k in 3 4 5 1; do python tools/overlay_build.py phase$k phase_cut:$k; done      # build container
  python tools/phase_profile.py OUT.md [WORKLOAD:AA ...]                             # GPU box

Phases of the cover kernel (cumulative, in execution order): 3 = loads + centred paths (P0 loads, P1a); 4 = + action
hit-test / move (P0); 5 = + task reward, termination, outputs (P0); 1 = + canvas edges (P1b); whole kernel = + coverage of
every batch and the run lists (P2).  Then the resample / fill kernel (P3, P4)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PHASES = (3, 4, 5, 1)
NAMES = {3: 'loads + centred paths', 4: '+ action/hit-test/move', 5: '+ task/termination', 1: '+ canvas edges (P1b)'}
N_ENVS = 8192


def measure(name, aa):
  import numpy as np
  import torch
  from spriteworld_amd import engine, workloads
  cfg, pool, sample = workloads.build(name, N_ENVS, episodes_per_env=4, seed=0, anti_aliasing=aa)
  eng = engine.Engine(cfg, pool, device=0)
  rng = np.random.default_rng(2000)
  acts = [torch.as_tensor(sample(rng), device=eng.device) for _ in range(8)]
  for i in range(6):
    eng.step(acts[i % 8])
  torch.cuda.synchronize()
  eng.timing(True)
  for i in range(40):
    eng.step(acts[i % 8])
  torch.cuda.synchronize()
  cover, second, n = eng.kernel_times_ms()
  eng.close()
  print('RESULT' + json.dumps({'cover_ms': cover / n, 'second_ms': second / n}))


def run(lib, name, aa):
  env = dict(os.environ)
  if lib:
    env['SWB_LIBRARY'] = lib
  out = subprocess.run([sys.executable, os.path.abspath(__file__), '--one', name, str(aa)], env=env, capture_output=True, text=True).stdout
  return json.loads([l for l in out.splitlines() if l.startswith('RESULT')][-1][6:])


def main():
  if sys.argv[1] == '--one':
    return measure(sys.argv[2], int(sys.argv[3]))
  out = sys.argv[1]
  workloads_ = [(w.split(':')[0], int(w.split(':')[1])) for w in sys.argv[2:]] or [('cluster_s5', 5), ('cluster_s5', 1), ('embodied_s12', 5)]
  lines = ['## Kernel time by phase (HIP events, %d envs, ms per launch; cover kernel cumulative with the increment in brackets)' % N_ENVS, '',
           '| workload | ' + ' | '.join('%d: %s' % (k, NAMES[k]) for k in PHASES) + ' | cover kernel = + coverage, run lists (P2) | resample / fill kernel (P3, P4) | step |',
           '|---|' + '---|' * (len(PHASES) + 3)]
  for name, aa in workloads_:
    ts = []
    for k in PHASES:
      lib = os.path.join(ROOT, 'spriteworld_amd', 'csrc', 'exp_phase%d.so' % k)
      ts.append(run(lib, name, aa)['cover_ms'] if os.path.exists(lib) else float('nan'))
    full = run(None, name, aa)
    ts.append(full['cover_ms'])
    cells = ['%.4f (+%.4f)' % (t, t - (ts[i - 1] if i else 0.0)) for i, t in enumerate(ts)]
    lines.append('| %s AA=%d | ' % (name, aa) + ' | '.join(cells) + ' | %.4f | %.4f |' % (full['second_ms'], full['cover_ms'] + full['second_ms']))
    print(lines[-1], flush=True)
  open(out, 'a').write('\n'.join(lines) + '\n\n')


if __name__ == '__main__':
  main()

#!/usr/bin/env python
"""The UNMODIFIED reference on every core of this host: BASELINE.md section 3 / SURVEY 8d's CPU baseline for the headline scene.

P worker processes (default os.cpu_count()), each owning ENVS/P reference `spriteworld.environment.Environment` objects built
from the reference's own generators -- the scene of BASELINE configs[2]: 5 sprites in 2 hue clusters (2 'blue' + 3 'green',
configs/cobra/clustering.py:41-46,71-109), SelectMove(0.25), Clustering(reward_range 10), 50-step episodes, 64x64 PILRenderer
with anti_aliasing 5 -- seeded np.random.seed(1000 + env index), stepped round-robin with actions from
np.random.RandomState(2000 + step).uniform(size=(ENVS, 4)) (every worker takes its rows).  Warm-up steps, then timed steps;
aggregate and per-core env-steps/s go to stdout as one JSON object (commit it under profiles/).

Runs wherever `oracle/ref_harness.py` finds the reference: /root/reference (the build container) or, on the GPU node, its
sourceless bytecode under oracle/_ref (oracle/stage_ref.py).  bench.py runs this script as a child process (a fresh
interpreter: no forking next to a HIP context) for its `cpu_baseline` block, kind "reference".  Every worker is ONE core:
the BLAS / OpenMP pools of numpy and sklearn are pinned to one thread.
usage: python tools/reference_cpu_baseline.py [ENVS] [STEPS] [WARMUP] [P]"""
import json
import os
for _v in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS', 'NUMEXPR_NUM_THREADS'):
  os.environ[_v] = '1'                     # before numpy loads: a worker process is one core
import multiprocessing as mp  # noqa: E402
import platform  # noqa: E402
import sys  # noqa: E402
import time  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_env(ref, index):
  import numpy as np
  from spriteworld import factor_distributions as distribs
  from spriteworld import sprite_generators, tasks
  from spriteworld.configs.cobra import common
  clusters = [distribs.Continuous('c0', 0.55, 0.65), distribs.Continuous('c0', 0.27, 0.37)]
  other = distribs.Product([
      distribs.Continuous('x', 0.1, 0.9), distribs.Continuous('y', 0.1, 0.9),
      distribs.Discrete('shape', ['square', 'triangle', 'circle']), distribs.Discrete('scale', [0.13]),
      distribs.Continuous('c1', 0.3, 1.), distribs.Continuous('c2', 0.9, 1.)])
  gens = [sprite_generators.generate_sprites(distribs.Product((other, c0)), num_sprites=n) for c0, n in zip(clusters, (2, 3))]
  gen = sprite_generators.shuffle(sprite_generators.chain_generators(*gens))
  np.random.seed(1000 + index)
  return ref.environment.Environment(task=tasks.Clustering(clusters, terminate_bonus=0., reward_range=10.),
                                     action_space=common.action_space(), renderers=common.renderers(), init_sprites=gen,
                                     max_episode_length=50)


def worker(rank, nproc, envs, steps, warmup, queue, barrier):
  import numpy as np
  from oracle import ref_harness
  ref = ref_harness.load_reference()
  mine = list(range(rank, envs, nproc))
  es = [make_env(ref, i) for i in mine]
  for e in es:
    e.reset()
  checksum = 0
  t0 = None
  for t in range(warmup + steps):
    if t == warmup:
      barrier.wait()
      t0 = time.perf_counter()
    acts = np.random.RandomState(2000 + t).uniform(size=(envs, 4))
    for e, i in zip(es, mine):
      try:
        ts = e.step(acts[i])
      except ZeroDivisionError:          # tasks.py:215 `1. / 0.` when every cluster has collapsed: the reference raises; start over
        ts = e.reset()
      checksum += int(ts.observation['image'][::8, ::8].sum())
  dt = time.perf_counter() - t0
  queue.put((rank, len(mine) * steps, dt, checksum))


def main():
  envs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
  steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
  warmup = int(sys.argv[3]) if len(sys.argv) > 3 else 20
  nproc = int(sys.argv[4]) if len(sys.argv) > 4 else (os.cpu_count() or 1)
  ctx = mp.get_context('fork')
  queue, barrier = ctx.Queue(), ctx.Barrier(nproc)
  procs = [ctx.Process(target=worker, args=(r, nproc, envs, steps, warmup, queue, barrier)) for r in range(nproc)]
  t0 = time.perf_counter()
  for p in procs:
    p.start()
  res = [queue.get() for _ in procs]
  for p in procs:
    p.join()
  wall = time.perf_counter() - t0
  total = sum(r[1] for r in res)
  slowest = max(r[2] for r in res)
  cpu = ''
  try:
    cpu = [l.split(':', 1)[1].strip() for l in open('/proc/cpuinfo') if l.startswith('model name')][0]
  except (OSError, IndexError):
    pass
  from oracle import ref_harness
  print(json.dumps({
      'what': 'unmodified reference (Python + PIL + matplotlib + sklearn), headline scene of BASELINE configs[2]',
      'reference_root': ref_harness.REFERENCE_ROOT, 'reference_kind': ref_harness.reference_kind(),
      'third_party': ref_harness.third_party_versions(),
      'where': os.environ.get('SWB_REF_WHERE', 'host ' + platform.node()), 'cpu': cpu, 'machine': platform.machine(),
      'host_cpus': os.cpu_count(),
      'processes': nproc, 'envs': envs, 'timed_steps_per_env': steps, 'warmup_steps_per_env': warmup,
      'env_steps_per_s_all_cores': total / slowest, 'env_steps_per_s_per_core': total / slowest / nproc,
      'timed_seconds_slowest_worker': slowest, 'wall_seconds_incl_setup': wall, 'frame_checksum': sum(r[3] for r in res)}, indent=1))


if __name__ == '__main__':
  main()

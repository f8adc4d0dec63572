#!/usr/bin/env python
"""CPU stress of the kernel SOURCE (the emulated build of tests/emu) against the oracle on configurations the fuzz workloads do not
reach: 8 .. 16 sprites, mostly non-convex shapes, scales up to 1.5 (sprites larger than the frame), positions off the frame, small
and extreme canvases, anti_aliasing 1 .. 8.  Per seed: 8 environments x 4 steps.  Outcomes: `ok` (everything bit-exact, no flag),
`flagged` (the engine flagged an environment SWB_ENV_ERR_SPAN_OVERFLOW -- a capacity limit, never silent: counted, and every
environment that is NOT flagged must still be exact), `MISMATCH` (a difference without a flag: a bug).
usage: python tools/stress_emu.py FIRST LAST [PROCESSES] [--dense | --row] [--tasks]
--dense: 16 spoked / starred sprites of scale 0.6 .. 1.0 piled on the middle of the frame (dozens of spans per canvas row: the
span lists' and run lists' capacities)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


DENSE = '--dense' in sys.argv
TASKS = '--tasks' in sys.argv      # Clustering (2 .. 8 clusters) / MetaAggregated / DragAndDrop / float64 positions / velocities too
ROW = '--row' in sys.argv          # 16 small spoked sprites side by side on the same canvas rows of a wide image: more visible spans
                                   # per row than the span lists / run lists hold -- the overflow must be FLAGGED, never silent


def build(seed, n_envs):
  from spriteworld_amd import action_spaces, lowering, renderers, synthetic, tasks
  r = np.random.default_rng(seed + 31337)
  w, h = (int(4 * r.integers(2, 33)) for _ in range(2))
  aa = int(r.integers(1, 9))
  while aa * w > 640 or aa * h > 1280:
    aa -= 1
  S = int(r.integers(8, 17))
  nonconvex = ['star_4', 'star_5', 'star_6', 'spoke_4', 'spoke_5', 'spoke_6']
  convex = ['triangle', 'square', 'pentagon', 'hexagon', 'octagon', 'circle']
  shape_names = tuple(r.choice(nonconvex, size=int(r.integers(1, 4)), replace=False)) + tuple(r.choice(convex, size=int(r.integers(0, 2)), replace=False))
  scales = tuple(float(v) for v in r.choice([0.05, 0.13, 0.3, 0.6, 1.0, 1.5], size=3))
  angles = tuple(int(v) for v in r.integers(0, 360, size=6))
  keep = bool(r.integers(0, 2))
  if DENSE:
    S = 16
    shape_names = tuple(r.choice(['spoke_6', 'star_6', 'spoke_5', 'star_5'], size=2, replace=False))
    scales = (0.6, 1.0, 0.8)
  if ROW:
    S, w, h, aa = 16, 256, int(4 * r.integers(2, 9)), 2
    shape_names = tuple(r.choice(['spoke_6', 'star_6', 'spoke_5', 'spoke_4'], size=2, replace=False))
    scales = (0.05, 0.06, 0.04)
  kind = int(r.integers(0, 3)) if TASKS else 0
  n_tasks, full = 1, False
  if kind == 0:
    task = tasks.FindGoalPosition(filter_distrib=None, goal_position=(0.5, 0.5), terminate_distance=float(r.uniform(0.02, 0.2)))
    labels = [[int(r.integers(0, 2))] for _ in range(S)]
  elif kind == 1:                                   # Clustering with 2 .. 8 clusters (Davies-Bouldin needs 2 <= k < members)
    k = int(r.integers(2, min(9, S // 2 + 1)))
    task = tasks.Clustering([None] * k, termination_threshold=float(r.uniform(1.0, 3.0)), terminate_bonus=float(r.integers(0, 2)),
                            sparse_reward=bool(r.integers(0, 2)), reward_range=float(r.integers(1, 12)))
    labels = [[c % k] for c in range(2 * k)] + [[int(r.integers(-1, k))] for _ in range(S - 2 * k)]
    full = True
  else:                                             # MetaAggregated over three goal tasks
    subs = [tasks.FindGoalPosition(filter_distrib=None, goal_position=(0.25 + 0.5 * (i % 2), 0.25 + 0.5 * (i // 2)),
                                   terminate_distance=0.2, raw_reward_multiplier=10.) for i in range(3)]
    task = tasks.MetaAggregated(subs, reward_aggregator=str(r.choice(['sum', 'max', 'min', 'mean'])),
                                termination_criterion=str(r.choice(['all', 'any'])), terminate_bonus=float(r.integers(0, 2)))
    labels = [[int(i % 3 == t) for t in range(3)] for i in range(S)]
    n_tasks = 3
  pick = int(r.integers(0, 3))
  aspace = (action_spaces.Embodied(step_size=0.1, motion_cost=float(r.choice([0.0, 0.4]))) if pick == 0 else
            action_spaces.SelectMove(scale=0.5, motion_cost=float(r.choice([0.0, 0.7]))) if pick == 1 or not TASKS else
            action_spaces.DragAndDrop(scale=0.5, motion_cost=float(r.choice([0.0, 1.3]))))
  rend = {'image': renderers.PILRenderer(image_size=(w, h), anti_aliasing=aa,
                                         bg_color=tuple(int(v) for v in (r.integers(0, 256, 3) * r.integers(0, 2))),
                                         color_to_rgb=renderers.hsv_to_rgb)}
  P = n_envs * 2
  pool = synthetic.make_pool(r, P, S, [(0.0, 1.0)] * S, labels, n_tasks=n_tasks, shape_names=shape_names, scales=scales, angles=angles,
                             xy_range=(0.3, 0.7) if DENSE else ((-0.3, 1.3) if not keep else (0.0, 1.0)))
  pool.n_sprites[:] = S if (DENSE or ROW) else np.maximum(r.integers(S // 2, S + 1, size=P), 1)
  if ROW:
    pool.y[:] = 0.5 + r.uniform(-0.02, 0.02, size=pool.y.shape).astype(np.float32)
    pool.x[:] = ((np.arange(S)[None, :] + 0.5) / S + r.uniform(-0.01, 0.01, size=pool.x.shape)).astype(np.float32)
  f32_pos = True
  if TASKS:
    if full:
      pool.n_sprites[:] = S
    f32_pos = bool(r.integers(0, 2))
    if not f32_pos:
      pool.x[:] = r.uniform(0.0, 1.0, size=pool.x.shape)
      pool.y[:] = r.uniform(0.0, 1.0, size=pool.y.shape)
    if r.integers(0, 2):
      vel = r.uniform(-0.03, 0.03, size=(2,) + pool.x.shape)
      if f32_pos:
        vel = vel.astype(np.float32).astype(np.float64)
      pool.x_vel[:], pool.y_vel[:] = vel[0], vel[1]
  pool.assign_round_robin(n_envs, 2)
  cfg = lowering.lower_config(task, aspace, rend, keep, 6, n_envs, S, f32_pos)

  def sample(rng):
    if cfg.action_space == 2:
      return np.stack([rng.integers(0, 2, n_envs), rng.integers(0, 4, n_envs)], axis=1).astype(np.int32)
    return rng.uniform(0, 1, (n_envs, 4))
  return cfg, pool, sample, dict(w=w, h=h, aa=aa, S=S, shapes=shape_names, scales=scales)


def one(seed):
  from oracle import oracle
  from tests import _emu_engine
  n_envs = 8
  try:
    cfg, pool, sample, desc = build(seed, n_envs)
    ora, eng = oracle.Engine(cfg, pool), _emu_engine.EmuEngine(cfg, pool)
    rng = np.random.default_rng(seed)
    flagged = np.zeros(n_envs, bool)
    for t in range(4):
      a = sample(rng)
      want = ora.step(a)
      eng.step(a)
      got = eng.outputs_host()
      flagged |= (got['error'] & 4) != 0
      other = (got['error'] & ~np.uint8(4)) != (want['error'] & ~np.uint8(4))
      ok = ~flagged
      st_g, st_o = eng.state(), ora.state()
      rg, ro = got['reward'][ok], want['reward'][ok]
      rew_bad = not (np.array_equal(np.isnan(rg), np.isnan(ro)) and np.array_equal(rg[~np.isnan(rg)].view(np.uint64), ro[~np.isnan(ro)].view(np.uint64)))
      bad = rew_bad or not np.array_equal(got['success'][ok], want['success'][ok]) or \
          other[ok].any() or not np.array_equal(got['obs'][ok], want['obs'][ok]) or \
          not np.array_equal(st_g['x'][ok].view(np.uint64), st_o['x'][ok].view(np.uint64)) or \
          not np.array_equal(got['step_type'][ok], want['step_type'][ok])
      if bad:
        eng.close()
        return seed, 'MISMATCH', desc
    eng.close()
    return seed, ('flagged %d' % flagged.sum()) if flagged.any() else 'ok', desc
  except Exception as e:  # pylint: disable=broad-except
    return seed, 'EXCEPTION %r' % (e,), None


def main():
  argv = [a for a in sys.argv if not a.startswith('--')]
  first, last = int(argv[1]), int(argv[2])
  procs = int(argv[3]) if len(argv) > 3 else 8
  from tests.emu import build_emu
  from oracle import oracle
  build_emu.build()
  oracle.build()
  import multiprocessing as mp
  counts = {}
  with mp.get_context('spawn').Pool(procs) as pool:
    for seed, what, desc in pool.imap_unordered(one, range(first, last)):
      key = what.split()[0]
      counts[key] = counts.get(key, 0) + 1
      if key != 'ok':
        print(seed, what, desc, flush=True)
  print('stress seeds [%d, %d):' % (first, last), counts, flush=True)


if __name__ == '__main__':
  main()

#!/usr/bin/env python
"""Generates tests/golden/*.npz from the UNMODIFIED reference (run in the build container only).

For each listed reference config: draw `n_eps` episodes with the reference's own
`init_sprites()` under a fixed seed, lower them with spriteworld_amd.lowering (pool arrays +
SwbConfig bytes), then run the reference `Environment` itself (from /root/reference, under the
shims of oracle/ref_harness.py) on a fixed action sequence, replaying those episodes, and record
what it returned at every step: step_type, reward, discount, success, sprite positions, the
CRC32 of every frame and the first frames in full.

The fixtures travel to the GPU box, where /root/reference does not exist: tests/test_golden.py
checks the oracle (CPU) and the HIP engine (GPU) against them.
Versions the fixtures were generated with are stored in each file (numpy / Pillow / matplotlib /
scikit-learn / glibc are un-pinned by the reference; the reference-as-run defines parity).
"""
import copy
import ctypes
import importlib
import os
import platform
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref_harness  # noqa: E402
from spriteworld_amd import lowering  # noqa: E402

CASES = [
    # name, module, mode, n_eps, n_steps, seed
    ('cobra_more_distractors', 'spriteworld.configs.cobra.goal_finding_more_distractors', 'train', 24, 160, 11),
    ('cobra_new_position_test', 'spriteworld.configs.cobra.goal_finding_new_position', 'test', 16, 120, 12),
    ('cobra_clustering', 'spriteworld.configs.cobra.clustering', 'train', 12, 200, 13),
    ('cobra_sorting', 'spriteworld.configs.cobra.sorting', 'train', 12, 200, 14),
    ('cobra_exploration', 'spriteworld.configs.cobra.exploration', 'train', 24, 200, 15),
    ('examples_embodied', 'spriteworld.configs.examples.goal_finding_embodied', 'train', 8, 200, 16),
    ('examples_goal_clustering', 'spriteworld.configs.examples.goal_finding_clustering', 'train', 8, 160, 17),
    # float32 action arrays (the dtype action_spec() declares), with a motion cost
    ('cobra_clustering_f32_actions', 'spriteworld.configs.cobra.clustering', 'test', 12, 200, 18, 'float32', 0.6),
    ('cobra_sorting_f32_actions', 'spriteworld.configs.cobra.sorting', 'test', 12, 160, 19, 'float32', 0.25),
    # the remaining config x mode combinations (tests/configs/configs_test.py grid)
    ('cobra_new_shape_train', 'spriteworld.configs.cobra.goal_finding_new_shape', 'train', 12, 120, 20),
    ('cobra_new_shape_test', 'spriteworld.configs.cobra.goal_finding_new_shape', 'test', 12, 120, 21),
    ('cobra_more_targets_train', 'spriteworld.configs.cobra.goal_finding_more_targets', 'train', 12, 120, 22),
    ('cobra_more_targets_test', 'spriteworld.configs.cobra.goal_finding_more_targets', 'test', 12, 120, 23),
    ('cobra_new_position_train', 'spriteworld.configs.cobra.goal_finding_new_position', 'train', 12, 120, 24),
    ('cobra_more_distractors_test', 'spriteworld.configs.cobra.goal_finding_more_distractors', 'test', 12, 120, 25),
    ('cobra_exploration_test', 'spriteworld.configs.cobra.exploration', 'test', 12, 120, 26),
    ('examples_embodied_test', 'spriteworld.configs.examples.goal_finding_embodied', 'test', 8, 120, 27),
    ('examples_goal_clustering_test', 'spriteworld.configs.examples.goal_finding_clustering', 'test', 8, 120, 28),
]
# round 6: tasks whose filters key on position (tests/_position_cases.py, built from the reference's own classes): name, float32 positions
POSITION_CASES = [(c, f32) for c in ('goal_x_lt_half', 'goal_two_xbands', 'goal_all_but_a_corner', 'goal_whole_frame_half_open',
                                     'goal_f64_bounds', 'meta_swap_sides') for f32 in (True, False)
                  if (c, f32) not in (('goal_two_xbands', False), ('goal_all_but_a_corner', True), ('goal_f64_bounds', False))]
FULL_FRAMES = 6


def versions():
  import PIL
  import matplotlib
  import sklearn
  return 'numpy %s; Pillow %s; matplotlib %s; scikit-learn %s; %s; python %s' % (
      np.__version__, PIL.__version__, matplotlib.__version__, sklearn.__version__,
      ' '.join(platform.libc_ver()), platform.python_version())


def make(name, module, mode, n_eps, n_steps, seed, action_dtype='float64', motion_cost=None, position_case=None):
  from spriteworld import action_spaces, environment
  from spriteworld import renderers as ref_renderers
  np.random.seed(seed)
  if position_case is not None:
    from tests import _position_cases as pc
    case, f32 = position_case
    ns = pc.namespace_of_reference()
    task, aspace, rends, keep, max_len = pc.environment_parts(ns, case)
    config = dict(task=task, action_space=aspace, renderers=rends, keep_in_frame=keep, max_episode_length=max_len)
    episodes = pc.episodes_of(ns, case, f32, n_episodes=n_eps, seed=seed)
    module, mode = 'tests/_position_cases.py:%s' % case, 'f32' if f32 else 'f64'
  else:
    mod = importlib.import_module(module)
    config = mod.get_config(mode)
    if motion_cost is not None:
      config['action_space'] = action_spaces.SelectMove(scale=0.25, motion_cost=motion_cost)
    gen = config['init_sprites']
    episodes = [gen() for _ in range(n_eps)]
  task, aspace, rends = config['task'], config['action_space'], config['renderers']
  S = max(len(e) for e in episodes)
  pos_dt = lowering.position_dtype(episodes)
  cfg = lowering.lower_config(task, aspace, rends, True, config['max_episode_length'], 1, S,
                              pos_is_f32=(pos_dt == np.float32), action_dtype=np.dtype(action_dtype))
  pool = lowering.lower_episodes(episodes, task, rends, max_sprites=S).assign_round_robin(1)
  # reference run, replaying the same episodes (the constructor draws once, environment.py:68)
  def fresh():     # the constructor's own draw, then the pool order (wrapping) as new objects each time
    yield copy.deepcopy(episodes[0])
    while True:
      for e in episodes:
        yield copy.deepcopy(e)
  it = fresh()
  config = dict(config)
  config['init_sprites'] = lambda: next(it)
  config['renderers'] = dict(rends)
  config['renderers']['success'] = ref_renderers.Success()
  env = environment.Environment(**config)
  rng = np.random.RandomState(seed + 1000)
  embodied = cfg.action_space == 2
  actions = np.zeros((n_steps, 2), np.int32) if embodied else np.zeros((n_steps, 4), np.dtype(action_dtype))
  out = dict(step_type=np.zeros(n_steps, np.uint8), reward=np.zeros(n_steps, np.float64),
             discount=np.zeros(n_steps, np.float32), success=np.zeros(n_steps, np.uint8),
             x=np.zeros((n_steps, S)), y=np.zeros((n_steps, S)), n_sprites=np.zeros(n_steps, np.int32),
             frame_crc=np.zeros(n_steps, np.uint32))
  frames = []
  for t in range(n_steps):
    if embodied:
      a = np.array([rng.randint(0, 2), rng.randint(0, 4)])
      ts = env.step([int(a[0]), int(a[1])])
    else:
      a = rng.uniform(0, 1, 4).astype(action_dtype)
      if position_case is not None and t % 3 == 0 and env._sprites:     # click ON a sprite: something moves most steps
        a[:2] = np.asarray(env._sprites[rng.randint(len(env._sprites))].position, dtype=np.float64)
      ts = env.step(a)
    actions[t] = a
    out['step_type'][t] = int(ts.step_type)
    out['reward'][t] = np.nan if ts.reward is None else float(ts.reward)
    out['discount'][t] = np.nan if ts.discount is None else float(ts.discount)
    out['success'][t] = bool(ts.observation['success'])
    pos = np.array([s.position for s in env._sprites], dtype=np.float64).reshape(-1, 2)
    out['n_sprites'][t] = len(pos)
    out['x'][t, :len(pos)] = pos[:, 0]
    out['y'][t, :len(pos)] = pos[:, 1]
    img = np.ascontiguousarray(ts.observation['image'])
    out['frame_crc'][t] = zlib.crc32(img.tobytes())
    if t < FULL_FRAMES:
      frames.append(img)
  save = {'cfg_bytes': np.frombuffer(bytes(cfg), dtype=np.uint8), 'actions': actions,
          'frames': np.stack(frames), 'versions': np.array(versions()),
          'source': np.array('%s mode=%s seed=%d' % (module, mode, seed))}
  for f in lowering.Pool.FIELDS:
    save['pool_' + f] = getattr(pool, f)
  if pool.cell_label is not None:
    save['pool_cell_label'] = pool.cell_label
  for k, v in out.items():
    save['ref_' + k] = v
  np.savez_compressed(os.path.join(HERE, name + '.npz'), **save)
  print(name, 'S=%d' % S, 'episodes seen: %d' % int((out['step_type'] == 0).sum()),
        'lasts: %d' % int((out['step_type'] == 2).sum()))


def main():
  ref_harness.load_reference()
  only = set(sys.argv[1:])          # optional: names of the cases to (re)generate
  for case in CASES:
    if not only or case[0] in only:
      make(*case)
  for k, (case, f32) in enumerate(POSITION_CASES):
    name = 'position_%s_%s' % (case, 'f32' if f32 else 'f64')
    if not only or name in only:
      make(name, None, None, 10, 150, 40 + k, position_case=(case, f32))
  # shape tables
  from spriteworld import constants
  import json
  with open(os.path.join(HERE, 'shapes.json'), 'w') as f:
    json.dump({k: [[float(a).hex(), float(b).hex()] for a, b in v] for k, v in constants.SHAPES.items()}, f,
              indent=0)
  print('wrote shapes.json')


if __name__ == '__main__':
  main()

"""The BASELINE.json workloads as concrete (config, pool, action sampler) triples.

Each workload is the batched form of a reference configuration (SURVEY.md
section 8d concretises BASELINE.json's `configs`):

  goal_s5      1024/8192 envs, 5 sprites (2 targets + 3 distractors),
               SelectMove(scale 0.25), FindGoalPosition(filter c0 in [0, 0.4),
               terminate_distance 0.075), 64x64 AA=5 HSV, episodes of 20 steps
               (reference: configs/cobra/goal_finding_more_distractors.py:54-96,
               configs/cobra/common.py:26-38)
  cluster_s5   8192 envs, 5 sprites in 2 hue clusters (2 'blue' + 3 'green'),
               Clustering(threshold 2.5, reward_range 10), episodes of 50 steps
               (reference: configs/cobra/clustering.py:41-46,71-109)   <- headline
  embodied_s12 8192 envs, 11 sprites + body circle (scale 0.07), Embodied(step
               0.05), FindGoalPosition, 128x128 AA=5
               (reference: configs/examples/goal_finding_embodied.py:47-118)
  sorting_s4   MetaAggregated of FindGoalPosition sub-tasks
               (reference: configs/cobra/sorting.py:41-140)

All use synthetic pools drawn with numpy (spriteworld_amd/synthetic.py).
"""
import numpy as np

from spriteworld_amd import action_spaces
from spriteworld_amd import lowering
from spriteworld_amd import renderers
from spriteworld_amd import synthetic
from spriteworld_amd import tasks


def _renderers(size, aa):
  return {'image': renderers.PILRenderer(image_size=(size, size), anti_aliasing=aa,
                                         color_to_rgb=renderers.hsv_to_rgb)}


def build(name, num_envs, episodes_per_env=4, seed=0, anti_aliasing=5):
  """Returns (SwbConfig, Pool, sample_actions(rng) -> ndarray).

  A name ending in `_f32a` is the same workload driven with float32 actions."""
  f32_actions = name.endswith('_f32a')
  if f32_actions:
    name = name[:-len('_f32a')]
  rng = np.random.default_rng(seed)
  P = num_envs * episodes_per_env
  aa = anti_aliasing
  if name == 'goal_s5':
    task = tasks.FindGoalPosition(filter_distrib=None, terminate_distance=0.075)
    aspace = action_spaces.SelectMove(scale=0.25)
    rend = _renderers(64, aa)
    hues = [(0.0, 0.4)] * 2 + [(0.5, 0.9)] * 3
    labels = [[1]] * 2 + [[0]] * 3
    pool = synthetic.make_pool(rng, P, 5, hues, labels)
    cfg = lowering.lower_config(task, aspace, rend, True, 20, num_envs, 5, True)
  elif name == 'cluster_s5':
    task = tasks.Clustering([None, None], terminate_bonus=0., reward_range=10.)
    aspace = action_spaces.SelectMove(scale=0.25)
    rend = _renderers(64, aa)
    hues = [(0.55, 0.65)] * 2 + [(0.27, 0.37)] * 3
    labels = [[0]] * 2 + [[1]] * 3
    pool = synthetic.make_pool(rng, P, 5, hues, labels)
    cfg = lowering.lower_config(task, aspace, rend, True, 50, num_envs, 5, True)
  elif name == 'cluster_s5_st':
    # the headline scene with squares and triangles only (episodes of at most 20 polygon vertices: a small LDS footprint --
    # used to measure the cover kernel at occupancies its usual footprint does not allow)
    task = tasks.Clustering([None, None], terminate_bonus=0., reward_range=10.)
    aspace = action_spaces.SelectMove(scale=0.25)
    rend = _renderers(64, aa)
    hues = [(0.55, 0.65)] * 2 + [(0.27, 0.37)] * 3
    labels = [[0]] * 2 + [[1]] * 3
    pool = synthetic.make_pool(rng, P, 5, hues, labels, shape_names=('square', 'triangle'))
    cfg = lowering.lower_config(task, aspace, rend, True, 50, num_envs, 5, True)
  elif name == 'cluster6_s12':
    # six clusters of two (a 6 x 6 Davies-Bouldin ratio matrix), float32 positions; one sprite in no cluster
    task = tasks.Clustering([None] * 6, termination_threshold=1.2, terminate_bonus=1., reward_range=6.)
    aspace = action_spaces.SelectMove(scale=0.25)
    rend = _renderers(64, aa)
    hues = [(0.08 * i, 0.08 * i + 0.05) for i in range(6) for _ in range(2)] + [(0.9, 1.0)]
    labels = [[i] for i in range(6) for _ in range(2)] + [[-1]]
    pool = synthetic.make_pool(rng, P, 13, hues, labels, scales=(0.06, 0.09))
    cfg = lowering.lower_config(task, aspace, rend, True, 30, num_envs, 13, True)
  elif name == 'cluster9_s16':
    # nine clusters (seven pairs, two singletons) of 16 sprites: a 9 x 9 Davies-Bouldin ratio matrix -- more rows than the
    # kernel's 64 lanes take in one pass (7 rows of 9) -- in float32 positions
    task = tasks.Clustering([None] * 9, termination_threshold=1.2, terminate_bonus=1., reward_range=6.)
    aspace = action_spaces.SelectMove(scale=0.25)
    rend = _renderers(64, aa)
    hues = [(0.08 * i, 0.08 * i + 0.05) for i in range(7) for _ in range(2)] + [(0.6, 0.65), (0.7, 0.75)]
    labels = [[i] for i in range(7) for _ in range(2)] + [[7], [8]]
    pool = synthetic.make_pool(rng, P, 16, hues, labels, scales=(0.06, 0.09))
    cfg = lowering.lower_config(task, aspace, rend, True, 30, num_envs, 16, True)
  elif name.startswith('geom_'):
    # image geometry sweep: geom_<W>x<H> (non-square, wide images; anti_aliasing from the argument)
    w, h = (int(v) for v in name[len('geom_'):].split('x'))
    task = tasks.FindGoalPosition(filter_distrib=None, terminate_distance=0.075)
    aspace = action_spaces.SelectMove(scale=0.3)
    rend = {'image': renderers.PILRenderer(image_size=(w, h), anti_aliasing=aa, bg_color=(7, 30, 110),
                                           color_to_rgb=renderers.hsv_to_rgb)}
    pool = synthetic.make_pool(rng, P, 4, [(0.0, 1.0)] * 4, [[1], [0], [1], [0]],
                               shape_names=('square', 'triangle', 'circle', 'star_5', 'spoke_4'),
                               scales=(0.1, 0.2, 0.4), angles=tuple(range(0, 360, 23)), xy_range=(0.0, 1.0))
    cfg = lowering.lower_config(task, aspace, rend, True, 12, num_envs, 4, True)
  elif name in ('ragged_s16', 'ragged_s16_embodied'):
    # episodes of 0..16 sprites (the engine's maximum), some without any target
    task = tasks.FindGoalPosition(filter_distrib=None, terminate_distance=0.15)
    aspace = (action_spaces.Embodied(step_size=0.1) if name.endswith('embodied')
              else action_spaces.SelectMove(scale=0.4))
    rend = _renderers(64, aa)
    labels = [[int(i % 3 == 0)] for i in range(16)]
    pool = synthetic.make_pool(rng, P, 16, [(0.0, 1.0)] * 16, labels,
                               shape_names=('square', 'triangle', 'circle', 'star_4'), scales=(0.08, 0.15))
    pool.n_sprites[:] = rng.integers(0, 17, size=P)
    pool.n_sprites[:4] = (0, 16, 1, 0)
    cfg = lowering.lower_config(task, aspace, rend, True, 8, num_envs, 16, True)
  elif name.startswith('fuzz_'):
    # randomised configuration (seeded by the name): image geometry, anti-aliasing, sprite counts,
    # shapes / scales / angles, task, action space, position dtype, velocities, background
    frng = np.random.default_rng(int(name[len('fuzz_'):]) + 977)
    w, h = (int(4 * frng.integers(4, 41)) for _ in range(2))
    aa = int(frng.choice([1, 2, 3, 5]))
    if aa * w > 640:
      aa = max(1, 640 // w)
    S = int(frng.integers(1, 11))
    all_shapes = list(renderers.__dict__.get('_unused', ())) or ['triangle', 'square', 'pentagon', 'hexagon', 'octagon',
                                                                 'circle', 'star_4', 'star_5', 'star_6', 'spoke_4',
                                                                 'spoke_5', 'spoke_6']
    shape_names = tuple(frng.choice(all_shapes, size=int(frng.integers(1, 6)), replace=False))
    scales = tuple(float(v) for v in frng.choice([0.02, 0.05, 0.08, 0.13, 0.2, 0.35, 0.6], size=3))
    angles = tuple(int(v) for v in frng.integers(0, 360, size=5)) + (0,)
    kind = int(frng.integers(0, 4))
    f32_pos = bool(frng.integers(0, 2))
    embodied = bool(frng.integers(0, 3) == 0)
    keep = bool(frng.integers(0, 2))
    n_tasks = 1
    if kind == 0 or S < 4:
      task = tasks.FindGoalPosition(filter_distrib=None, goal_position=(float(frng.uniform(0.2, 0.8)), float(frng.uniform(0.2, 0.8))),
                                    terminate_distance=float(frng.uniform(0.05, 0.3)), terminate_bonus=float(frng.integers(0, 3)),
                                    weights_dimensions=(float(frng.integers(1, 3)), float(frng.integers(0, 3))),
                                    sparse_reward=bool(frng.integers(0, 2)), raw_reward_multiplier=float(frng.integers(1, 60)))
      labels = [[int(frng.integers(0, 2))] for _ in range(S)]
    elif kind in (1, 2):
      k = 2 if S < 6 else int(frng.integers(2, 4))
      task = tasks.Clustering([None] * k, termination_threshold=float(frng.uniform(1.0, 3.0)),
                              terminate_bonus=float(frng.integers(0, 2)), sparse_reward=bool(frng.integers(0, 2)),
                              reward_range=float(frng.integers(1, 12)))
      labels = [[c % k] for c in range(2 * k)] + [[int(frng.integers(-1, k))] for _ in range(S - 2 * k)]
    else:
      subs = [tasks.FindGoalPosition(filter_distrib=None, goal_position=(0.25 + 0.5 * (i % 2), 0.25 + 0.5 * (i // 2)),
                                     terminate_distance=0.2, raw_reward_multiplier=10.) for i in range(3)]
      task = tasks.MetaAggregated(subs, reward_aggregator=str(frng.choice(['sum', 'max', 'min', 'mean'])),
                                  termination_criterion=str(frng.choice(['all', 'any'])), terminate_bonus=float(frng.integers(0, 2)))
      labels = [[int(i % 3 == t) for t in range(3)] for i in range(S)]
      n_tasks = 3
    if embodied:
      aspace = action_spaces.Embodied(step_size=float(frng.choice([0.05, 0.1])), motion_cost=float(frng.choice([0.0, 0.4])))
    elif frng.integers(0, 2):
      aspace = action_spaces.SelectMove(scale=float(frng.choice([0.25, 0.5])), motion_cost=float(frng.choice([0.0, 0.7])))
    else:
      aspace = action_spaces.DragAndDrop(scale=float(frng.choice([0.25, 0.5])), motion_cost=float(frng.choice([0.0, 1.3])))
    rend = {'image': renderers.PILRenderer(image_size=(w, h), anti_aliasing=aa,
                                           bg_color=tuple(int(v) for v in (frng.integers(0, 256, 3) * frng.integers(0, 2))),
                                           color_to_rgb=renderers.hsv_to_rgb)}
    pool = synthetic.make_pool(rng, P, S, [(0.0, 1.0)] * S, labels, n_tasks=n_tasks, shape_names=shape_names, scales=scales,
                               angles=angles, xy_range=(-0.05, 1.05) if not keep else (0.0, 1.0))
    if not f32_pos:
      pool.x[:] = rng.uniform(0.0, 1.0, size=pool.x.shape)
      pool.y[:] = rng.uniform(0.0, 1.0, size=pool.y.shape)
    if frng.integers(0, 2):
      vel = rng.uniform(-0.02, 0.02, size=(2,) + pool.x.shape)
      if f32_pos:
        vel = vel.astype(np.float32).astype(np.float64)
      pool.x_vel[:], pool.y_vel[:] = vel[0], vel[1]
    pool.n_sprites[:] = np.maximum(rng.integers(S // 2, S + 1, size=P), 1 if embodied else 0)
    if kind in (1, 2) and S >= 4:
      pool.n_sprites[:] = S            # keep Davies-Bouldin's label-count precondition (2 <= k < m)
    cfg = lowering.lower_config(task, aspace, rend, keep, int(frng.integers(3, 15)), num_envs, S, f32_pos)
  elif name == 'embodied_s12':
    task = tasks.FindGoalPosition(filter_distrib=None, terminate_distance=0.075)
    aspace = action_spaces.Embodied(step_size=0.05)
    rend = _renderers(128, aa)
    hues = [(0.0, 0.4)] * 4 + [(0.5, 0.9)] * 7
    labels = [[1]] * 4 + [[0]] * 7
    pool = synthetic.make_pool(
        rng, P, 11, hues, labels, shape_names=('square', 'triangle', 'circle', 'star_5', 'spoke_4'),
        scales=(0.13, 0.2, 0.3), angles=(0, 17, 45, 90), xy_range=(0.2, 0.8),
        body=dict(scale=0.07, hue=(0.0, 1.0), shape='circle', label=0))
    cfg = lowering.lower_config(task, aspace, rend, True, 50, num_envs, 12, True)
  elif name == 'sorting_s4':
    goals = [(0.75, 0.75), (0.75, 0.25), (0.25, 0.75), (0.25, 0.25)]
    subs = [tasks.FindGoalPosition(filter_distrib=None, goal_position=g, terminate_distance=0.075,
                                   raw_reward_multiplier=20.) for g in goals]
    task = tasks.MetaAggregated(subs, reward_aggregator='sum', termination_criterion='all')
    aspace = action_spaces.SelectMove(scale=0.25)
    rend = _renderers(64, aa)
    hues = [(0.9, 1.0), (0.55, 0.65), (0.27, 0.37), (0.73, 0.83)]
    labels = [[int(i == j) for j in range(4)] for i in range(4)]
    pool = synthetic.make_pool(rng, P, 4, hues, labels, n_tasks=4)
    cfg = lowering.lower_config(task, aspace, rend, True, 50, num_envs, 4, True)
  elif name == 'tiny_s6':
    # sprites of a few canvas pixels: vertices collapse onto shared integer points (degenerate
    # edges, duplicate vertices), which takes the general branch of the corner rule
    task = tasks.FindGoalPosition(filter_distrib=None, terminate_distance=0.075)
    aspace = action_spaces.SelectMove(scale=0.5)
    rend = _renderers(64, aa)
    pool = synthetic.make_pool(
        rng, P, 6, [(0.0, 1.0)] * 6, [[1]] * 6,
        shape_names=('circle', 'star_6', 'spoke_6', 'octagon', 'spoke_5', 'star_5'),
        scales=(0.004, 0.01, 0.02, 0.035, 0.05), angles=tuple(range(0, 360, 13)), xy_range=(0.0, 1.0))
    cfg = lowering.lower_config(task, aspace, rend, True, 20, num_envs, 6, True)
  elif name == 'wide_s4':
    # very large, rotated, partly off-canvas sprites: exercises x-chunked scan conversion, clipping
    task = tasks.FindGoalPosition(filter_distrib=None, terminate_distance=0.075)
    aspace = action_spaces.SelectMove(scale=0.5)
    rend = _renderers(64, aa)
    pool = synthetic.make_pool(
        rng, P, 4, [(0.0, 1.0)] * 4, [[1]] * 4,
        shape_names=('square', 'triangle', 'circle', 'star_4', 'spoke_5', 'pentagon'),
        scales=(0.5, 0.7, 0.95, 1.3), angles=tuple(range(0, 360, 11)), xy_range=(0.0, 1.0))
    cfg = lowering.lower_config(task, aspace, rend, True, 20, num_envs, 4, True)
  elif name in ('f64_drag', 'f64_cluster'):
    # test-style sprites: float64 positions, rotations, non-convex shapes, velocities, no clipping
    rend = {'image': renderers.PILRenderer(image_size=(32, 32), anti_aliasing=aa if aa != 5 else 3,
                                           bg_color=(10, 200, 30))}
    if name == 'f64_drag':
      task = tasks.FindGoalPosition(goal_position=(0.3, 0.6), terminate_distance=0.1,
                                    terminate_bonus=5., weights_dimensions=(1, 3),
                                    raw_reward_multiplier=7)
      aspace = action_spaces.DragAndDrop(scale=0.5, motion_cost=1.3)
      labels = [[1], [1], [0], [1]]
    else:
      task = tasks.Clustering([None, None, None], termination_threshold=1.5, terminate_bonus=2.,
                              reward_range=4.)
      aspace = action_spaces.SelectMove(scale=0.3, motion_cost=0.7)
      labels = [[0], [1], [2], [0], [1], [-1]]
    ns = len(labels)
    pool = synthetic.make_pool(
        rng, P, ns, [(0.0, 1.0)] * ns, labels,
        shape_names=('star_5', 'spoke_4', 'hexagon', 'triangle', 'star_6', 'spoke_6', 'octagon'),
        scales=(0.1, 0.2, 0.35), angles=tuple(range(0, 360, 7)), xy_range=(0.1, 0.9))
    # float64 positions with full mantissas and small velocities
    pool.x[:] = rng.uniform(0.1, 0.9, size=pool.x.shape)
    pool.y[:] = rng.uniform(0.1, 0.9, size=pool.y.shape)
    pool.x_vel[:] = rng.uniform(-0.02, 0.02, size=pool.x.shape)
    pool.y_vel[:] = rng.uniform(-0.02, 0.02, size=pool.x.shape)
    cfg = lowering.lower_config(task, aspace, rend, False, 15, num_envs, ns, False)
  else:
    raise ValueError('unknown workload ' + name)
  pool.assign_round_robin(num_envs, episodes_per_env)
  if f32_actions and cfg.action_space != 2:
    cfg.action_is_f32 = 1

  if cfg.action_space == 2:
    def sample(r):
      return np.stack([r.integers(0, 2, num_envs), r.integers(0, 4, num_envs)], 1).astype(np.int32)
  else:
    def sample(r):
      a = r.uniform(0.0, 1.0, size=(num_envs, 4))
      return a.astype(np.float32) if cfg.action_is_f32 else a
  return cfg, pool, sample


def build_protocol_8d(name, num_envs, episodes_per_env=4, anti_aliasing=5, env_offset=0, total_envs=None):
  """SURVEY.md section 8d's input protocol, literally, for the scenes of BASELINE configs[1] / configs[2]: every environment's
  reset pool is drawn by the reference's OWN generators -- this package's mirrors of `sprite_generators` / `factor_distributions`,
  which consume numpy's global stream draw for draw like the reference's (tests/test_host_api.py) -- under
  `np.random.seed(1000 + env_index)`; actions of step t are `np.random.RandomState(2000 + t).uniform(size=(N, 4))`.
  A shard of a larger job passes the index of its first environment and the job's size (`env_offset`, `total_envs`): seeds and
  action rows are those of the global environment indices.
  Returns (SwbConfig, Pool, actions_of_step(t) -> f64[N, 4]).  (`build()` draws the same distributions with numpy directly, in
  a fraction of the time: tests use that.)"""
  from spriteworld_amd import factor_distributions as distribs
  from spriteworld_amd import sprite_generators
  rend = _renderers(64, anti_aliasing)
  common = [distribs.Continuous('x', 0.1, 0.9), distribs.Continuous('y', 0.1, 0.9),
            distribs.Discrete('shape', ['square', 'triangle', 'circle']), distribs.Discrete('scale', [0.13]),
            distribs.Continuous('c1', 0.3, 1.), distribs.Continuous('c2', 0.9, 1.)]
  if name == 'cluster_s5':       # configs/cobra/clustering.py:41-46,71-109: 2 'blue' + 3 'green', shuffled
    clusters = [distribs.Continuous('c0', 0.55, 0.65), distribs.Continuous('c0', 0.27, 0.37)]
    gens = [sprite_generators.generate_sprites(distribs.Product(common + [c0]), num_sprites=n) for c0, n in zip(clusters, (2, 3))]
    task = tasks.Clustering(clusters, terminate_bonus=0., reward_range=10.)
    max_len = 50
  elif name == 'goal_s5':        # configs/cobra/goal_finding_more_distractors.py:54-96: 2 targets + 3 distractors
    hues = [distribs.Continuous('c0', 0., 0.4), distribs.Continuous('c0', 0.5, 0.9)]
    gens = [sprite_generators.generate_sprites(distribs.Product(common + [c0]), num_sprites=n) for c0, n in zip(hues, (2, 3))]
    task = tasks.FindGoalPosition(filter_distrib=hues[0], terminate_distance=0.075)
    max_len = 20
  else:
    raise ValueError('no section-8d protocol for workload ' + name)
  gen = sprite_generators.shuffle(sprite_generators.chain_generators(*gens))
  state = np.random.get_state()                     # (the caller's global stream is left as it was)
  try:
    episodes = []
    for env in range(num_envs):
      np.random.seed(1000 + env_offset + env)
      episodes.extend(gen() for _ in range(episodes_per_env))
  finally:
    np.random.set_state(state)
  aspace = action_spaces.SelectMove(scale=0.25)
  cfg = lowering.lower_config(task, aspace, rend, True, max_len, num_envs, 5, True)
  pool = lowering.lower_episodes(episodes, task, rend, max_sprites=5)
  pool.assign_round_robin(num_envs, episodes_per_env)

  total = int(total_envs or (env_offset + num_envs))

  def actions_of_step(t):
    return np.ascontiguousarray(np.random.RandomState(2000 + int(t)).uniform(size=(total, 4))[env_offset:env_offset + num_envs])
  return cfg, pool, actions_of_step

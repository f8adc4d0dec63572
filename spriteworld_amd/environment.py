"""Batched Spriteworld environment on MI355X behind the reference's dm_env surface.

`BatchedEnvironment(task, action_space, renderers, init_sprites, keep_in_frame,
max_episode_length, metadata, num_envs=N)` takes exactly the keyword arguments of
the reference's `Environment` (reference: spriteworld/environment.py:34-72), i.e.
the dict a config's `get_config()` returns, plus `num_envs`.  `reset()` /
`step(actions)` / `observation_spec()` / `action_spec()` / `.action_space` follow
:74-161 with a leading N axis on every array:

  step_type u8[N] (0 FIRST, 1 MID, 2 LAST), reward f64[N] (NaN where the
  reference returns None, i.e. on FIRST steps), discount f32[N] (NaN on FIRST),
  observation {name: tensor}: the PILRenderer's key -> u8[N, H, W, 3] (device
  tensor), a `Success` renderer's key -> bool[N].

Per-environment auto-reset is the reference's: the step after a LAST step
ignores its action and returns FIRST (:90-91).  Where new episodes come from:

  * `device_reset='auto'` (default): when `init_sprites` was built with
    sprite_generators.* from Product / SetMinus / Continuous / Discrete
    distributions (every shipped config), episodes are drawn ON THE DEVICE
    (device_sampler, Philox streams keyed by `seed`) and the idle pool entries
    are redrawn every `refresh_every` steps, so an environment never meets an
    episode twice -- the reference's "fresh init_sprites() at every reset";
  * otherwise (opaque `init_sprites` callables) the pool holds
    `episodes_per_env` host draws per environment and each environment CYCLES
    through them: a long run replays these episodes until `refill_pool()` is
    called (which restarts every environment).  Use a declarative generator, or
    raise `episodes_per_env`, for training runs.

The tensors of a returned time step (observation, reward, ...) are the engine's
output buffers: the next `step()` overwrites them in place; clone what must
outlive a step (e.g. `obs` vs `next_obs`).

`Environment` is the N = 1 form returning `dm_env.TimeStep`s of numpy values, a
drop-in for the reference class in `example_run_loop.py:62-80`.
"""
import collections

import numpy as np
import torch

from spriteworld_amd import _abi
from spriteworld_amd import device_sampler
from spriteworld_amd import dm_env_compat as dm_env
from spriteworld_amd import engine as _engine
from spriteworld_amd import lowering

BatchedTimeStep = collections.namedtuple('BatchedTimeStep',
                                         ['step_type', 'reward', 'discount', 'observation'])


class EnvironmentError_(RuntimeError):
  pass


class BatchedEnvironment(object):
  """N independent Spriteworld environments stepped by the HIP engine (two kernels per step, csrc/swb_kernels.hip.inc)."""

  def __init__(self, task, action_space, renderers, init_sprites, keep_in_frame=True,
               max_episode_length=1000, metadata=None, num_envs=1, episodes_per_env=8,
               max_sprites=None, device=0, check_errors=32, action_dtype=np.float64,
               global_env_offset=0, device_reset='auto', refresh_every='auto', seed=None):
    self._task = task
    self._action_space = action_space
    self._renderers = renderers
    self._init_sprites = init_sprites
    self._keep_in_frame = keep_in_frame
    self._max_episode_length = max_episode_length
    self._metadata = metadata
    self._num_envs = int(num_envs)
    self._episodes_per_env = int(episodes_per_env)
    # index of this shard's first environment in the whole job (distributed.shard_range): device-side
    # reset sampling draws entry streams by global index, so shards never repeat each other's episodes
    self._global_env_offset = int(global_env_offset)
    # Per-environment error flags (conditions that make the reference raise) are read back every
    # `check_errors` steps (True/1: every step, which costs a device sync per step; 0/False: never;
    # `check()` can be called at any time).
    self._check_errors = int(check_errors)
    self._steps_since_check = 0
    # With device-side reset sampling: redraw the idle pool entries every `refresh_every` steps (0: never;
    # refresh_pool() can be called by hand).  episodes_per_env * shortest episode length is a safe period.
    # 'auto': 2 * (episodes_per_env - 1) steps -- an episode lasts at least two steps (FIRST + one more), so no
    # environment can run out of unseen entries between two refreshes.
    self._refresh_every = 2 * (self._episodes_per_env - 1) if refresh_every == 'auto' else int(refresh_every)
    self._steps_since_refresh = 0
    self._image_key, self._pil = lowering.find_pil_renderer(renderers)
    self._success_keys = [k for k, r in renderers.items() if type(r).__name__ == 'Success']
    self._factor_keys = {k: r for k, r in renderers.items() if type(r).__name__ == 'SpriteFactors'}
    unsupported = [k for k, r in renderers.items()
                   if k != self._image_key and k not in self._success_keys and k not in self._factor_keys]
    if unsupported:
      raise lowering.LoweringError('renderers not supported on device: %s' % unsupported)
    # Episodes are drawn on the device when init_sprites is a DeviceSampler, or -- device_reset=True /
    # 'auto' -- when it was built with sprite_generators.* from Product/SetMinus/Continuous/Discrete
    # distributions (the generator's closures are read, device_sampler.from_generator); 'auto' falls
    # back to calling init_sprites() on the host when that is not possible.
    self._sampler = init_sprites if isinstance(init_sprites, device_sampler.DeviceSampler) else None
    if self._sampler is None and device_reset:
      try:
        sampler = device_sampler.from_generator(init_sprites, seed=0)
        sampler.lower(task, renderers)
        # `seed` keys the Philox episode streams; None draws it from numpy's global stream, so np.random.seed()
        # makes runs reproducible and different seeds give different episodes.  Drawn only once the generator is
        # known to lower: a generator that falls back to the host path sees numpy's stream untouched.
        sampler.seed = int(np.random.randint(0, 2**31 - 1)) if seed is None else int(seed)
        self._sampler = sampler
      except lowering.LoweringError:
        if device_reset != 'auto':
          raise
    if self._sampler is not None and self._episodes_per_env < 2 and refresh_every == 'auto':
      import warnings
      warnings.warn('device-side reset sampling with episodes_per_env=1 cannot refresh the pool while it is in use: every '
                    'environment replays its one episode (raise episodes_per_env, or call refill_pool() yourself)')
    if self._sampler is not None:   # episodes are drawn by the engine (swb_sample_pool)
      episodes = None
      S = max_sprites or max(self._sampler.max_sprites, 1)
      pos_dt = np.float32
    else:
      episodes = self._draw_episodes()
      S = max_sprites or max([len(ep) for ep in episodes] + [1])
      pos_dt = lowering.position_dtype(episodes)
    self._max_sprites = S
    # SelectMove noise (action_spaces.py:69-75) is added to the action tensor on the device.  `action +
    # np.random.normal(...)` is float64 whatever the action's dtype (array + float64 array), so a noisy action
    # space always runs the engine's float64-action arithmetic.
    ns = getattr(action_space, '_noise_scale', None)
    if ns:
      action_dtype = np.float64
    self._cfg = lowering.lower_config(task, action_space, renderers, keep_in_frame,
                                      max_episode_length, self._num_envs, S,
                                      pos_is_f32=(pos_dt == np.float32), action_dtype=action_dtype)
    self._noise_scale = None if not ns else torch.as_tensor(np.asarray(ns, dtype=np.float64))
    self._noise_gen = None
    if self._sampler is not None:
      self._sampler_spec = self._sampler.lower(task, renderers)
      self._engine = _engine.Engine(self._cfg, None, device=device)
      self.refill_pool()
    else:
      pool = lowering.lower_episodes(episodes, task, renderers, max_sprites=S)
      pool.assign_round_robin(self._num_envs, self._episodes_per_env)
      self._engine = _engine.Engine(self._cfg, pool, device=device)
    self._render = self._pil is not None
    self._attr_assigned = {}      # (env, sprite, 'angle' | 'scale') -> (episode, the assigned value was an np.float32)

  # ------------------------------------------------------------------ pool
  def _draw_episodes(self):
    return [list(self._init_sprites()) for _ in range(self._num_envs * self._episodes_per_env)]

  def refill_pool(self):
    """Draws a fresh pool with init_sprites(); every environment restarts (next step is FIRST).

    With a DeviceSampler the pool is drawn by a HIP kernel (no host sampling, no upload)."""
    if self._sampler is not None:
      k = self._episodes_per_env
      base = np.arange(self._num_envs, dtype=np.int32) * k
      self._engine.sample_pool(self._sampler_spec, self._num_envs * k, base,
                               np.full(self._num_envs, k, np.int32), self._sampler.next_seed(),
                               first_entry=self._global_env_offset * k)
      return
    episodes = self._draw_episodes()
    pool = lowering.lower_episodes(episodes, self._task, self._renderers,
                                   max_sprites=self._max_sprites)
    pool.assign_round_robin(self._num_envs, self._episodes_per_env)
    self._engine.set_pool(pool)

  def refresh_pool(self):
    """DeviceSampler only: redraws every pool entry that no environment is playing right now, without
    resetting anything.  Called every few episodes, environments never meet an episode twice."""
    if self._sampler is None:
      raise lowering.LoweringError('refresh_pool() needs init_sprites to be a device_sampler.DeviceSampler')
    self._engine.resample_pool(self._sampler.next_seed(),
                               first_entry=self._global_env_offset * self._episodes_per_env)

  # ------------------------------------------------------------------ dm_env surface
  @property
  def num_envs(self):
    return self._num_envs

  @property
  def action_space(self):
    return self._action_space

  @property
  def engine(self):
    return self._engine

  def action_spec(self):
    return self._action_space.action_spec()

  def observation_spec(self):
    spec = {}
    if self._image_key is not None:
      spec[self._image_key] = self._pil.observation_spec()
    for k in self._success_keys:
      spec[k] = dm_env.specs.Array(shape=(), dtype=np.bool_)
    for k, r in self._factor_keys.items():
      spec[k] = dm_env.specs.Array(shape=(self._max_sprites, len(r._factors)), dtype=np.float64)
    return spec

  def _timestep(self):
    e = self._engine
    if self._check_errors:
      self._steps_since_check += 1
      if self._steps_since_check >= self._check_errors:
        self.check()
    obs = {}
    if self._image_key is not None:
      obs[self._image_key] = e.obs
    for k in self._success_keys:
      obs[k] = e.success.bool()
    self._add_factor_obs(obs)
    return BatchedTimeStep(e.step_type, e.reward, e.discount, obs)

  def _add_factor_obs(self, obs):
    if not self._factor_keys:
      return
    from spriteworld_amd import sprite as sprite_lib
    full = self._engine.factors()
    for k, r in self._factor_keys.items():
      cols = [sprite_lib.FACTOR_NAMES.index(f) for f in r._factors]
      obs[k] = full if cols == list(range(10)) else full[:, :, cols]

  def seed_noise(self, seed):
    """Seeds the generator of the SelectMove action noise."""
    self._noise_gen = torch.Generator(device=self._engine.device)
    self._noise_gen.manual_seed(int(seed))

  def check(self):
    """Raises what the reference would have raised if any environment flagged an error."""
    self._steps_since_check = 0
    err = int(self._engine.error.max().item())
    if err:
      self._engine.error.zero_()          # the flags are sticky on the device (include/swb.h): consumed here
      if err & _abi.ENV_ERR_DB_ZERO:
        raise ZeroDivisionError('float division by zero (Davies-Bouldin score is 0)')
      if err & _abi.ENV_ERR_DB_LABELS:
        raise ValueError('Number of labels is invalid for davies_bouldin_score')
      raise EnvironmentError_('internal engine error bits 0x%x' % err)

  def reset(self):
    """Environment.reset() for every environment: returns the FIRST time steps."""
    self._engine.reset_all()
    return self.step(self.null_actions())

  def null_actions(self):
    if self._cfg.action_space == _abi.ACTION_EMBODIED:
      return torch.zeros((self._num_envs, 2), dtype=torch.int32, device=self._engine.device)
    dt = torch.float32 if self._cfg.action_is_f32 else torch.float64
    return torch.zeros((self._num_envs, 4), dtype=dt, device=self._engine.device)

  def step(self, actions):
    """actions: [N, 4] float (SelectMove / DragAndDrop) or [N, 2] int (Embodied)."""
    if self._noise_scale is not None:
      e = self._engine
      if not isinstance(actions, torch.Tensor):
        actions = torch.as_tensor(np.ascontiguousarray(actions))
      actions = actions.to(device=e.device)
      noise = torch.randn(actions.shape, dtype=torch.float64, device=e.device, generator=self._noise_gen)
      actions = actions.to(torch.float64) + noise * self._noise_scale.to(e.device)      # float64 from here on (see __init__)
    self._engine.step(actions, render=self._render)
    if self._refresh_every and self._sampler is not None:
      self._steps_since_refresh += 1
      if self._steps_since_refresh >= self._refresh_every:
        self._steps_since_refresh = 0
        self.refresh_pool()
    return self._timestep()

  def observation(self):
    """environment.py:136-142: every renderer's view of the sprites AS THEY ARE NOW -- the frame is rendered now (swb_render)
    and a Success renderer evaluates the task now (swb_evaluate; environment.py:128-131 `state()` calls `success()` on the
    current sprites), so that a setter or `set_positions` since the last step shows in both (round-5 advice: the Success key
    used to carry the last step's flag)."""
    obs = {}
    if self._image_key is not None:
      obs[self._image_key] = self._engine.render()
    if self._success_keys:
      success = self._engine.evaluate().bool()
      for k in self._success_keys:
        obs[k] = success
    self._add_factor_obs(obs)
    return obs

  def state(self):
    """Live structure-of-arrays state (host copies): x, y [N, S], n_sprites, step_count, ..."""
    return self._engine.state()

  def sample_actions(self):
    return self._action_space.sample(self._num_envs)

  # ------------------------------------------------------------------ live sprites (sprite.py:152-175)
  def sprites(self, env=0):
    """The sprites of environment `env` in its current episode, back to front, as `sprite.LiveSprite` handles
    (the reference's `env.state()['sprites']`): `env.sprites(3)[0].angle = 45` acts on the device state."""
    from spriteworld_amd import sprite as sprite_lib
    n = self._engine.env_state(env)['n_sprites']
    return [sprite_lib.LiveSprite(self, env, k) for k in range(n)]

  def set_sprite_attr(self, env, sprite, name, value):
    """`sprites(env)[sprite].<name> = value` for name in ('shape', 'angle', 'scale'): the reference's setters
    (sprite.py:152-175) through swb_set_sprite_attr.  The sprite's task labels are re-evaluated with the new
    factor (tasks.py:134-137,196-205), so a filter keyed on shape / angle / scale follows the change."""
    from spriteworld_amd import shapes as shapes_lib
    from spriteworld_amd import sprite as sprite_lib
    attr = {'shape': _abi.ATTR_SHAPE, 'angle': _abi.ATTR_ANGLE, 'scale': _abi.ATTR_SCALE}[name]
    live = sprite_lib.LiveSprite(self, env, sprite)
    factors = live.factors
    before = factors[name]               # (one read of the sprite serves the labels and the setter's difference)
    factors[name] = value
    proxy = collections.namedtuple('_Factors', ['factors'])(factors)
    subs = lowering.subtasks_of(self._task)
    label = np.array([lowering._label_of(sub, proxy) for sub in subs], dtype=np.int8)  # pylint: disable=protected-access
    # tasks whose filters key on position: the sprite's label in every cell of the task's grid, with the new attribute
    cell_label = None
    f32 = bool(self._cfg.pos_is_f32)
    cuts = [lowering.position_cuts(sub, f32) for sub in subs]
    if any(xc or yc for xc, yc in cuts):
      cell_label = np.zeros((len(subs), _abi.SWB_MAX_CELLS), np.int8)
      for t, (sub, (xc, yc)) in enumerate(zip(subs, cuts)):
        if xc or yc:
          cell_label[t] = lowering.cell_labels_of(sub, proxy, xc, yc, np.float32 if f32 else np.float64)
    delta = None
    episode = self._engine.env_state(env)['episode']
    if name != 'shape':
      # sprite.py:163,173 take `a - self._angle` / `s - self._scale` with whatever types the two carry: an np.float32
      # attribute (factor distributions draw float32) makes it a float32 subtraction under NEP 50.  numpy decides here too.
      old = before
      old = np.float32(old) if self._attr_is_f32(env, sprite, name, episode, old) else float(old)
      delta = float(value - old)
    self._engine.set_sprite_attr(env, sprite, attr, shapes_lib.shape_index(value) if name == 'shape' else float(value),
                                 delta=delta, label=label, cell_label=cell_label)
    if name != 'shape':           # from now on the attribute is the object the caller assigned
      self._attr_assigned[(env, sprite, name)] = (episode, getattr(value, 'dtype', None) == np.float32)

  def _attr_is_f32(self, env, sprite, name, episode, value):
    """Is this sprite's angle / scale an np.float32 in the reference?  A value assigned through a setter keeps the type it
    was given; otherwise it is what the episode's generator drew -- factor distributions draw float32, Discrete candidates
    and Sprite() arguments are Python numbers -- which the pool records per sprite (swb_pool::attr_f32: written by
    `lowering.lower_episodes` from the sprite objects and by the device sampler from the factor's kind)."""
    del value
    rec = self._attr_assigned.get((env, sprite, name))
    if rec is not None and rec[0] == episode:
      return rec[1]
    angle_f32, scale_f32 = self._engine.sprite_types(env, sprite)
    return angle_f32 if name == 'angle' else scale_f32

  def close(self):
    self._engine.close()


class EnvironmentGroups(object):
  """`num_groups` independent BatchedEnvironments of `num_envs // num_groups` environments, each on its own
  HIP stream: the double-buffered stepping of RL samplers (the policy works on one group's observations while
  another group steps).  Work of different groups is unordered, so consecutive steps of different groups
  overlap on the GPU and hide each launch's fill and drain (bench.py `extra`, `*_2_groups_2_streams`).  Each group's `global_env_offset` is set, so device-side reset sampling draws the episodes of
  one `num_envs` batch."""

  def __init__(self, num_groups=2, num_envs=2, device=0, **kwargs):
    if num_envs % num_groups:
      raise ValueError('num_envs must be a multiple of num_groups')
    per = num_envs // num_groups
    base = int(kwargs.pop('global_env_offset', 0))
    self.streams = [torch.cuda.Stream(device=torch.device('cuda', device)) for _ in range(num_groups)]
    self.groups = []
    for g in range(num_groups):
      with torch.cuda.stream(self.streams[g]):
        self.groups.append(BatchedEnvironment(num_envs=per, device=device, global_env_offset=base + g * per, **kwargs))

  def __len__(self):
    return len(self.groups)

  def reset(self, g):
    with torch.cuda.stream(self.streams[g]):
      return self.groups[g].reset()

  def step(self, g, actions):
    """Steps group `g` on its stream; the returned tensors are valid on `self.streams[g]`."""
    with torch.cuda.stream(self.streams[g]):
      return self.groups[g].step(actions)

  def synchronize(self):
    for s in self.streams:
      s.synchronize()

  def close(self):
    self.synchronize()
    for env in self.groups:
      env.close()


class Environment(object):
  """Single environment with the reference's exact return types (dm_env.TimeStep of numpy).

  `device_reset` defaults to False here -- unlike BatchedEnvironment's 'auto' -- on purpose: the N = 1 form exists to be
  swapped for the reference class (example_run_loop.py:62-80), and calling `init_sprites()` on the host at every reset keeps
  the episodes those of the reference under the same np.random.seed().

  The episode sequence is the reference's: the k-th call of `init_sprites()` plays the same part in both.  Call 0 -- pool
  entry 0 -- is the sprites the reference's CONSTRUCTOR draws (`self._sprites = self._init_sprites()`, environment.py:68):
  what `state()`, `observation()`, `success()` ... see before the first step, never an episode that is stepped; the first
  `reset()` / `step()` plays call 1, the next one call 2, ...  Whether the caller looks at the state first changes nothing
  (the pool, `episodes_per_pool` calls, is drawn up front and again when it is used up)."""

  def __init__(self, task, action_space, renderers, init_sprites, keep_in_frame=True,
               max_episode_length=1000, metadata=None, episodes_per_pool=32, device=0,
               action_dtype=np.float64, device_reset=False):
    self._batched = BatchedEnvironment(
        task, action_space, renderers, init_sprites, keep_in_frame=keep_in_frame,
        max_episode_length=max_episode_length, metadata=metadata, num_envs=1,
        episodes_per_env=episodes_per_pool, device=device, action_dtype=action_dtype, check_errors=1,
        device_reset=device_reset)
    self._episodes_per_pool = episodes_per_pool
    self._episodes_used = 0
    self._reset_next_step = True
    self._started = False           # the constructor's sprites (pool entry 0) are on the device

  @property
  def action_space(self):
    return self._batched.action_space

  def action_spec(self):
    return self._batched.action_spec()

  def observation_spec(self):
    return self._batched.observation_spec()

  def reward_spec(self):
    """dm_env.Environment's default (the reference does not override it): a float scalar."""
    return dm_env.specs.Array(shape=(), dtype=float, name='reward')

  def discount_spec(self):
    """dm_env.Environment's default: a float scalar in [0, 1]."""
    return dm_env.specs.BoundedArray(shape=(), dtype=float, minimum=0., maximum=1., name='discount')

  def _convert(self, ts):
    step_type = dm_env.StepType(int(ts.step_type[0].item()))
    obs = {}
    for k, v in ts.observation.items():
      a = v[0].cpu().numpy()
      obs[k] = bool(a) if a.shape == () else a
    if step_type == dm_env.StepType.FIRST:
      return dm_env.TimeStep(step_type, None, None, obs)
    return dm_env.TimeStep(step_type, float(ts.reward[0].item()), float(ts.discount[0].item()), obs)

  def _maybe_refill(self):
    # a fresh draw of init_sprites() per episode, like the reference; the pool is re-drawn when used up
    if self._episodes_used >= self._episodes_per_pool:
      self._batched.refill_pool()
      self._episodes_used = 0
    self._episodes_used += 1

  def reset(self):
    self._ensure_started()
    self._maybe_refill()
    ts = self._convert(self._batched.reset())
    self._reset_next_step = False
    return ts

  def step(self, action):
    if self._reset_next_step:
      return self.reset()
    a = np.asarray(action)[None]
    ts = self._convert(self._batched.step(a))
    if ts.last():
      self._reset_next_step = True
    return ts

  # ---- the rest of the reference's public surface (environment.py:80-86,110-142), acting on the device state
  def _ensure_started(self):
    """The reference's constructor already holds sprites (`self._sprites = self._init_sprites()`, environment.py:68), so
    its introspection methods work before the first step: pool entry 0 is put on the device for them, once, by the first
    call of anything -- `reset()` included, so the episodes that are stepped start with entry 1 whether or not anybody
    looked at the constructor's sprites.  The first `step()` still resets (environment.py:90-91), as the reference does."""
    if not self._started:
      self._started = True
      self._maybe_refill()
      self._batched.reset()

  def success(self):
    """environment.py:80-81: `task.success(sprites)` of the sprites as they are NOW -- evaluated on the device (swb_evaluate:
    the cover kernel's task phase, no time step), so that a setter or `set_positions` since the last step is seen, as the
    reference sees it."""
    self._ensure_started()
    return bool(self._batched.engine.evaluate()[0].item())

  def should_terminate(self):
    """environment.py:83-86."""
    self._ensure_started()
    timeout = self._batched.engine.env_state(0)['step_count'] >= self._batched._max_episode_length  # pylint: disable=protected-access
    out_of_frame = any([s.out_of_frame for s in self.sprites])
    return self.success() or out_of_frame or timeout

  def sample_contained_position(self):
    """environment.py:110-126: a random sprite (np.random.randint), then a random position inside it."""
    sprites = self.sprites
    return sprites[np.random.randint(len(sprites))].sample_contained_position()

  def observation(self):
    """environment.py:136-142: every renderer's view of the current state, rendered now (swb_render: the cover and
    resample kernels without a state change), as numpy values."""
    self._ensure_started()
    obs = {}
    for k, v in self._batched.observation().items():
      a = v[0].cpu().numpy()
      obs[k] = bool(a) if a.shape == () else a
    return obs

  def state(self):
    """environment.py:128-134: {'sprites': [...], 'global_state': {'success': ..., 'metadata': ...}} -- the sprites as
    `LiveSprite` handles on the device state (attributes read and, for shape / angle / scale, assigned like the reference's).
    The structure-of-arrays state (x, y, step_count, ...) is `soa_state()`."""
    global_state = {'success': self.success()}
    if self._batched._metadata:  # pylint: disable=protected-access
      global_state['metadata'] = self._batched._metadata  # pylint: disable=protected-access
    return {'sprites': self.sprites, 'global_state': global_state}

  def soa_state(self):
    """Host copies of the live structure-of-arrays state (swb_get_state): x, y f64[1, S], n_sprites, step_count, ..."""
    return self._batched.state()

  @property
  def sprites(self):
    """The current episode's sprites (reference: `env.state()['sprites']`) as `sprite.LiveSprite` handles whose
    shape / angle / scale can be assigned like on the reference's Sprite (sprite.py:152-175)."""
    self._ensure_started()
    return self._batched.sprites(0)

  def close(self):
    self._batched.close()

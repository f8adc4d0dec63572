"""Oracle-backed stand-in for `spriteworld_amd.engine.Engine` (TEST INFRASTRUCTURE ONLY).

Lets the CPU suite exercise the host-side API above the C ABI (`environment.BatchedEnvironment`, `sprite.LiveSprite`)
and dry-run the GPU test scripts where no GPU exists.  It proves nothing about the kernels: on the GPU box the same
scripts run against the real engine (tests/test_gpu_setters.py).  Never imported by the product package.
"""
import numpy as np
import torch

from oracle import oracle


class FakeEngineError(RuntimeError):
  pass


class FakeEngine(object):

  def __init__(self, cfg, pool, device=0):
    self.cfg, self.pool = cfg, pool
    self.device = torch.device('cpu')
    self.N, self.S = cfg.n_envs, cfg.max_sprites
    self.obs_shape = (cfg.image_w, cfg.image_h, 3)
    self._ora = oracle.Engine(cfg, pool)
    self.obs = torch.zeros((self.N,) + self.obs_shape, dtype=torch.uint8)
    self.reward = torch.zeros(self.N, dtype=torch.float64)
    self.discount = torch.zeros(self.N, dtype=torch.float32)
    self.step_type = torch.zeros(self.N, dtype=torch.uint8)
    self.success = torch.zeros(self.N, dtype=torch.uint8)
    self.error = torch.zeros(self.N, dtype=torch.uint8)

  def close(self):
    pass

  def set_pool(self, pool):
    self.__init__(self.cfg, pool)

  def reset_all(self):
    self._ora.reset_all()

  def step(self, actions, render=True):
    if isinstance(actions, torch.Tensor):
      actions = actions.cpu().numpy()
    out = self._ora.step(actions, render=render)
    for k in ('reward', 'discount', 'step_type', 'success'):
      getattr(self, k).copy_(torch.from_numpy(out[k]))
    self.error |= torch.from_numpy(out['error'])
    if render:
      self.obs.copy_(torch.from_numpy(out['obs']))

  def render(self):
    self.obs.copy_(torch.from_numpy(self._ora.render()))
    return self.obs

  def evaluate(self):
    self.success.copy_(torch.from_numpy(self._ora.evaluate()))
    return self.success

  def outputs_host(self):
    return {k: getattr(self, k).numpy().copy() for k in ('obs', 'reward', 'discount', 'step_type', 'success', 'error')}

  def state(self):
    return self._ora.state()

  def env_state(self, env):
    st = self._ora.state()
    return {k: int(st[k][env]) for k in ('n_sprites', 'pool_entry', 'step_count', 'episode', 'reset_next')}

  def sprite_types(self, env, sprite):
    f = int(self.pool.attr_f32[self.env_state(env)['pool_entry'], sprite]) if self.pool is not None else 0
    return bool(f & 1), bool(f & 2)

  def set_positions(self, x, y):
    self._ora.set_positions(x, y)

  def set_sprite_attr(self, env, sprite, attr, value, delta=None, label=None, cell_label=None):
    try:
      self._ora.set_sprite_attr(env, sprite, attr, value, delta=delta, label=label, cell_label=cell_label)
    except ValueError as e:
      raise FakeEngineError(str(e))

  def get_sprite(self, env, sprite):
    return self._ora.get_sprite(env, sprite)

  def factors(self):
    st = self.state()
    out = np.zeros((self.N, self.S, 10))
    for n in range(self.N):
      e = st['pool_entry'][n]
      for s in range(st['n_sprites'][n]):
        sp = self.get_sprite(n, s)
        out[n, s] = [st['x'][n, s], st['y'][n, s], sp['shape'] + 1, sp['angle'], sp['scale']] + \
            list(self.pool.color[e, s]) + [self.pool.x_vel[e, s], self.pool.y_vel[e, s]]
    return torch.from_numpy(out)

  def variant(self):
    return {'kernel': 'swb_resample_kernel<oracle>', 'build_id': 'oracle'}

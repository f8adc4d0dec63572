"""Thin Python handle on the HIP engine: device buffers are torch tensors.

PyTorch is used here for device memory and streams only; all computation is in
libswb.so.  One `Engine` = one `swb_handle` = N environments on one GPU.
"""
import ctypes as C

import numpy as np
import torch

from spriteworld_amd import _abi
from spriteworld_amd import _lib
from spriteworld_amd import lanczos
from spriteworld_amd import shapes as _shapes


def _ptr(a):
  return None if a is None else C.c_void_p(a.ctypes.data)


class Engine(object):
  """N batched environments on one MI355X (`cfg`: _abi.SwbConfig, `pool`: lowering.Pool)."""

  def __init__(self, cfg, pool, device=0):
    if not torch.cuda.is_available():
      raise _lib.SwbError('no GPU visible: the Spriteworld engine has no CPU path')
    self.lib = _lib.load()
    self.cfg = cfg
    self.device = torch.device('cuda', device)
    self.N, self.S = cfg.n_envs, cfg.max_sprites
    # the reference's np.array(image) is [image_size[1], image_size[0], 3]
    self.obs_shape = (cfg.image_w, cfg.image_h, 3)
    h = C.c_void_p()
    _lib.check(self.lib.swb_create(C.byref(cfg), device, C.byref(h)))
    self._h = h
    verts, offs = _shapes.packed_table()
    _lib.check(self.lib.swb_upload_shapes(self._h, _ptr(verts), _ptr(offs), len(offs) - 1))
    aa = cfg.anti_aliasing
    if aa != 1:
      for axis, out_size in ((0, cfg.image_h), (1, cfg.image_w)):
        bounds, coeffs = lanczos.resample_tables(aa * out_size, out_size)
        bounds = np.ascontiguousarray(bounds)
        coeffs = np.ascontiguousarray(coeffs)
        _lib.check(self.lib.swb_upload_resample(self._h, axis, out_size, coeffs.shape[1],
                                                _ptr(bounds), _ptr(coeffs)))
    self.pool = None
    if pool is not None:
      self.set_pool(pool)
    with torch.cuda.device(self.device):
      self.obs = torch.zeros((self.N,) + self.obs_shape, dtype=torch.uint8, device=self.device)
      self.reward = torch.zeros(self.N, dtype=torch.float64, device=self.device)
      self.discount = torch.zeros(self.N, dtype=torch.float32, device=self.device)
      self.step_type = torch.zeros(self.N, dtype=torch.uint8, device=self.device)
      self.success = torch.zeros(self.N, dtype=torch.uint8, device=self.device)
      self.error = torch.zeros(self.N, dtype=torch.uint8, device=self.device)
    self._outs = self._make_outs(render=True)
    self._outs_norender = self._make_outs(render=False)
    self._rendered = 0        # rendering launches so far (the run lists are trimmed after TRIM_AFTER of them)

  TRIM_AFTER = 3

  def _make_outs(self, render):
    o = _abi.SwbOutputs()
    o.obs = self.obs.data_ptr() if render else None
    o.reward = self.reward.data_ptr()
    o.discount = self.discount.data_ptr()
    o.step_type = self.step_type.data_ptr()
    o.success = self.success.data_ptr()
    o.error = self.error.data_ptr()
    return o

  def close(self):
    if getattr(self, '_h', None):
      torch.cuda.synchronize(self.device)
      self.lib.swb_destroy(self._h)
      self._h = None

  def __del__(self):
    try:
      self.close()
    except Exception:  # pylint: disable=broad-except
      pass

  def _stream(self):
    return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

  def set_pool(self, pool):
    self.pool = pool
    cpool = pool.as_struct()
    _lib.check(self.lib.swb_set_pool(self._h, C.byref(cpool)))
    self._rendered = 0        # (a new pool restores the lists' full reservation: trimmed again after TRIM_AFTER launches)

  def sample_pool(self, spec, n_entries, pool_base, pool_len, seed, first_entry=0):
    """Draws `n_entries` episodes on the device from an _abi.SwbSampler (swb_sample_pool)."""
    base = np.ascontiguousarray(pool_base, dtype=np.int32)
    length = np.ascontiguousarray(pool_len, dtype=np.int32)
    assert base.shape == (self.N,) and length.shape == (self.N,)
    _lib.check(self.lib.swb_sample_pool(self._h, C.byref(spec), int(n_entries), _ptr(base), _ptr(length),
                                        C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), C.c_uint64(int(first_entry)),
                                        self._stream()))
    self.pool = None
    self._pool_entries = int(n_entries)
    self._rendered = 0

  def resample_pool(self, seed, first_entry=0):
    """Fresh episodes in every pool entry no environment is playing; nothing is reset (swb_resample_pool)."""
    _lib.check(self.lib.swb_resample_pool(self._h, C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF),
                                          C.c_uint64(int(first_entry)), self._stream()))

  def get_pool(self):
    """Host copy (lowering.Pool) of the pool the device currently holds."""
    from spriteworld_amd import lowering
    n = self.pool.n_entries if self.pool is not None else self._pool_entries
    pool = lowering.Pool(n, self.S, self.cfg.n_tasks)
    pool.pool_base = np.zeros(self.N, np.int32)
    pool.pool_len = np.zeros(self.N, np.int32)
    cpool = pool.as_struct()
    _lib.check(self.lib.swb_get_pool(self._h, C.byref(cpool)))
    return pool

  def reset_all(self):
    _lib.check(self.lib.swb_reset_all(self._h, self._stream()))

  def step(self, actions, render=True):
    """actions: device tensor f64[N,4] (f32 if cfg.action_is_f32) or i32[N,2] (Embodied)."""
    if self.cfg.action_space == _abi.ACTION_EMBODIED:
      want = torch.int32
    else:
      want = torch.float32 if self.cfg.action_is_f32 else torch.float64
    if not isinstance(actions, torch.Tensor):
      actions = torch.as_tensor(np.ascontiguousarray(actions), device=self.device)
    if actions.dtype != want or actions.device != self.device or not actions.is_contiguous():
      actions = actions.to(device=self.device, dtype=want).contiguous()
    assert actions.numel() == self.N * (2 if want == torch.int32 else 4), actions.shape
    self._last_actions = actions  # keep alive until the launch is consumed
    outs = self._outs if render else self._outs_norender
    _lib.check(self.lib.swb_step(self._h, C.c_void_p(actions.data_ptr()), C.byref(outs),
                                 self._stream()))
    if render:
      self._rendered += 1
      if self._rendered == self.TRIM_AFTER:
        self.trim()

  def trim(self):
    """Cuts the hand-off lists between the two kernels of a step from their start-up reservation (any scene of convex sprites:
    133 KB per environment on 12 sprites at 128x128) down to 1.25 x what the launches so far needed, plus a shared arena for the
    lists that outgrow that (swb_trim_run_lists; blocking, once -- `step()` calls it after its third rendering launch, so it
    falls into any warm-up).  Returns the units of 8 bytes a list owns afterwards.  Results never change."""
    if not hasattr(self.lib, 'swb_trim_run_lists'):      # (an A/B build of an older revision)
      return None
    cap = C.c_int32(0)
    _lib.check(self.lib.swb_trim_run_lists(self._h, C.byref(cap), self._stream()))
    return cap.value

  def render(self):
    _lib.check(self.lib.swb_render(self._h, C.c_void_p(self.obs.data_ptr()), self._stream()))
    return self.obs

  def evaluate(self):
    """task.success() of the sprites as they are now (environment.py:80-81), into `self.success`: after sprite setters or
    set_positions the flag of the last step no longer describes them."""
    _lib.check(self.lib.swb_evaluate(self._h, C.c_void_p(self.success.data_ptr()), self._stream()))
    return self.success

  def factors(self):
    """SpriteFactors observation: f64 [N, S, 10] device tensor (FACTOR_NAMES order, shape as ShapeType id)."""
    if getattr(self, '_factors', None) is None:
      self._factors = torch.zeros((self.N, self.S, 10), dtype=torch.float64, device=self.device)
    _lib.check(self.lib.swb_factors(self._h, C.c_void_p(self._factors.data_ptr()), self._stream()))
    return self._factors

  def state(self):
    st = {
        'x': np.zeros((self.N, self.S)), 'y': np.zeros((self.N, self.S)),
        'n_sprites': np.zeros(self.N, np.int32), 'pool_entry': np.zeros(self.N, np.int32),
        'step_count': np.zeros(self.N, np.int32), 'reset_next': np.zeros(self.N, np.uint8),
        'episode': np.zeros(self.N, np.int32),
    }
    cs = _abi.SwbState(*[a.ctypes.data for a in (st['x'], st['y'], st['n_sprites'], st['pool_entry'],
                                                  st['step_count'], st['reset_next'], st['episode'])])
    _lib.check(self.lib.swb_get_state(self._h, C.byref(cs), self._stream()))
    return st

  def env_state(self, env):
    """dict(n_sprites, pool_entry, step_count, episode, reset_next) of ONE environment (swb_get_env_state: a few bytes,
    whatever the batch size)."""
    out = np.zeros(5, np.int32)
    _lib.check(self.lib.swb_get_env_state(self._h, int(env), _ptr(out), self._stream()))
    return dict(zip(('n_sprites', 'pool_entry', 'step_count', 'episode', 'reset_next'), (int(v) for v in out)))

  def sprite_types(self, env, sprite):
    """(angle is np.float32, scale is np.float32) for a sprite of the episode `env` is playing: the types the reference's
    Sprite holds (swb_pool::attr_f32, recorded by lowering / the device sampler)."""
    f = C.c_int32(0)
    _lib.check(self.lib.swb_get_sprite_types(self._h, int(env), int(sprite), C.byref(f), self._stream()))
    return bool(f.value & 1), bool(f.value & 2)

  def set_positions(self, x, y):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    _lib.check(self.lib.swb_set_positions(self._h, _ptr(x), _ptr(y), self._stream()))

  def set_sprite_attr(self, env, sprite, attr, value, delta=None, label=None, cell_label=None):
    """sprite.py:152-175 setters on a live sprite (swb_set_sprite_attr; attr: _abi.ATTR_SHAPE / ATTR_ANGLE / ATTR_SCALE).
    cell_label: i8[n_tasks, SWB_MAX_CELLS], the sprite's labels per cell for tasks that key on position."""
    d = None if delta is None else C.byref(C.c_double(float(delta)))
    lab = None if label is None else np.ascontiguousarray(label, dtype=np.int8)
    _lib.check(self.lib.swb_set_sprite_attr(self._h, int(env), int(sprite), int(attr), float(value), d, _ptr(lab),
                                            self._stream()))
    if cell_label is not None:
      cells = np.ascontiguousarray(cell_label, dtype=np.int8)
      assert cells.shape == (self.cfg.n_tasks, _abi.SWB_MAX_CELLS), cells.shape
      _lib.check(self.lib.swb_set_sprite_cell_labels(self._h, int(env), int(sprite), _ptr(cells), self._stream()))

  def get_sprite(self, env, sprite):
    """dict(shape=index, angle, scale, path=f64[n,2]): the sprite as the engine currently sees it (swb_get_sprite)."""
    shape, nv = C.c_int32(0), C.c_int32(0)
    angle, scale = C.c_double(0.0), C.c_double(0.0)
    path = np.zeros((_abi.SWB_MAX_SHAPE_VERTS, 2), dtype=np.float64)
    _lib.check(self.lib.swb_get_sprite(self._h, int(env), int(sprite), C.byref(shape), C.byref(angle), C.byref(scale),
                                       C.byref(nv), _ptr(path), self._stream()))
    return {'shape': shape.value, 'angle': angle.value, 'scale': scale.value, 'path': path[:nv.value].copy()}

  def outputs_host(self):
    torch.cuda.synchronize(self.device)
    return {
        'obs': self.obs.cpu().numpy(), 'reward': self.reward.cpu().numpy(),
        'discount': self.discount.cpu().numpy(), 'step_type': self.step_type.cpu().numpy(),
        'success': self.success.cpu().numpy(), 'error': self.error.cpu().numpy(),
    }

  def variant(self):
    """dict(nw, ncol, vs, lds_bytes_per_wave, waves_per_simd, kernel, build_id): the step kernel this engine launches."""
    info = _abi.SwbVariantInfo()
    _lib.check(self.lib.swb_variant(self._h, C.byref(info)))
    d = {k: getattr(info, k) for k, _ in _abi.SwbVariantInfo._fields_}
    d['cover_kernel'] = 'swb_cover_kernel<%d>' % info.nw
    d['kernel'] = ('swb_resample_kernel<%d>' % info.vs if info.vs else
                   ('none (the cover kernel paints the frame)' if info.paint_in_cover else 'swb_fill_kernel'))
    d['build_id'] = self.lib.swb_build_id().decode()
    return d

  def timing(self, enable):
    """Three HIP events per step (before the cover kernel, between the two kernels, after the second): `step_time_ms()` /
    `kernel_times_ms()` read them.  A diagnostic: each event is a completion signal the device has to raise between two
    kernels that would otherwise follow each other directly -- measured, a run of back-to-back steps is 6 % slower with
    them (tools/exp_timing_overhead.py), so time a run with ONE pair of events around it and use this mode for the split."""
    _lib.check(self.lib.swb_timing_enable(self._h, int(enable)))

  def step_time_ms(self):
    ms, n = C.c_double(0.0), C.c_int64(0)
    _lib.check(self.lib.swb_step_time_ms(self._h, C.byref(ms), C.byref(n)))
    return ms.value, n.value

  def kernel_times_ms(self):
    """(cover ms, resample / fill ms, launches) since timing(True): the step interval split between its two kernels."""
    a, b, n = C.c_double(0.0), C.c_double(0.0), C.c_int64(0)
    _lib.check(self.lib.swb_kernel_times_ms(self._h, C.byref(a), C.byref(b), C.byref(n)))
    return a.value, b.value, n.value

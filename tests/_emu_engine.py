"""`engine.Engine`'s method surface over tests/emu/_build/libswb_emu.so (TEST INFRASTRUCTURE ONLY).

libswb_emu.so is the source of spriteworld_amd/csrc -- the C-ABI host side AND the fused step kernel -- compiled for the
host against an emulation of the HIP runtime and of the wave-level builtins (tests/emu): every work-item is a fibre,
cross-lane operations are rendezvous of the 64 lanes.  "Device" buffers are numpy arrays.  It exists so that the CPU
test suite can execute the kernel source itself against the oracle where no GPU is available; the product never
loads it (spriteworld_amd/_lib.py loads csrc/libswb.so, built by hipcc for gfx950, and raises when it is missing).
"""
import ctypes as C
import os

import numpy as np

from spriteworld_amd import _abi
from spriteworld_amd import lanczos
from spriteworld_amd import shapes as _shapes
from tests.emu import build_emu

_lib = None


class EmuError(RuntimeError):
  pass


def lib():
  global _lib
  if _lib is None:
    l = C.CDLL(build_emu.build())
    l.swb_last_error.restype = C.c_char_p
    l.swb_create.argtypes = [C.POINTER(_abi.SwbConfig), C.c_int, C.POINTER(C.c_void_p)]
    l.swb_destroy.argtypes = [C.c_void_p]
    l.swb_upload_shapes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
    l.swb_upload_resample.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    l.swb_set_pool.argtypes = [C.c_void_p, C.POINTER(_abi.SwbPool)]
    l.swb_sample_pool.argtypes = [C.c_void_p, C.POINTER(_abi.SwbSampler), C.c_int32, C.c_void_p, C.c_void_p, C.c_uint64,
                                  C.c_uint64, C.c_void_p]
    l.swb_resample_pool.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
    l.swb_get_pool.argtypes = [C.c_void_p, C.POINTER(_abi.SwbPool)]
    l.swb_reset_all.argtypes = [C.c_void_p, C.c_void_p]
    l.swb_step.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(_abi.SwbOutputs), C.c_void_p]
    l.swb_render.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    if hasattr(l, 'swb_evaluate') or not os.environ.get('SWB_EMU_CSRC'):      # (an older copy of the sources may lack it)
      l.swb_evaluate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    l.swb_factors.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    if hasattr(l, 'swb_trim_run_lists') or not os.environ.get('SWB_EMU_CSRC'):
      l.swb_trim_run_lists.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_void_p]
      l.swb_set_sprite_cell_labels.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    l.swb_get_state.argtypes = [C.c_void_p, C.POINTER(_abi.SwbState), C.c_void_p]
    l.swb_set_positions.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    l.swb_set_sprite_attr.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
    l.swb_get_sprite.argtypes = [C.c_void_p, C.c_int32, C.c_int32] + [C.c_void_p] * 6
    l.swb_variant.argtypes = [C.c_void_p, C.POINTER(_abi.SwbVariantInfo)]
    l.swb_build_id.restype = C.c_char_p
    _lib = l
  return _lib


def check(rc):
  if rc != 0:
    raise EmuError('swb error %d: %s' % (rc, lib().swb_last_error().decode()))


def _ptr(a):
  return None if a is None else C.c_void_p(a.ctypes.data)


class EmuEngine(object):
  """N environments stepped by the emulated kernel (`cfg`: _abi.SwbConfig, `pool`: lowering.Pool or None)."""

  def __init__(self, cfg, pool, device=0):
    self.lib = lib()
    self.cfg = cfg
    self.N, self.S = cfg.n_envs, cfg.max_sprites
    self.obs_shape = (cfg.image_w, cfg.image_h, 3)
    h = C.c_void_p()
    check(self.lib.swb_create(C.byref(cfg), 0, C.byref(h)))
    self._h = h
    verts, offs = _shapes.packed_table()
    check(self.lib.swb_upload_shapes(self._h, _ptr(verts), _ptr(offs), len(offs) - 1))
    aa = cfg.anti_aliasing
    if aa != 1:
      for axis, out_size in ((0, cfg.image_h), (1, cfg.image_w)):
        bounds, coeffs = lanczos.resample_tables(aa * out_size, out_size)
        bounds, coeffs = np.ascontiguousarray(bounds), np.ascontiguousarray(coeffs)
        check(self.lib.swb_upload_resample(self._h, axis, out_size, coeffs.shape[1], _ptr(bounds), _ptr(coeffs)))
    self.pool = None
    if pool is not None:
      self.set_pool(pool)
    self.obs = np.full((self.N,) + self.obs_shape, 0x5A, dtype=np.uint8)      # garbage: every byte must be written
    self.reward = np.zeros(self.N, dtype=np.float64)
    self.discount = np.zeros(self.N, dtype=np.float32)
    self.step_type = np.zeros(self.N, dtype=np.uint8)
    self.success = np.zeros(self.N, dtype=np.uint8)
    self.error = np.zeros(self.N, dtype=np.uint8)

  def _outs(self, render):
    o = _abi.SwbOutputs()
    o.obs = self.obs.ctypes.data if render else None
    o.reward, o.discount = self.reward.ctypes.data, self.discount.ctypes.data
    o.step_type, o.success, o.error = self.step_type.ctypes.data, self.success.ctypes.data, self.error.ctypes.data
    return o

  def close(self):
    if getattr(self, '_h', None):
      self.lib.swb_destroy(self._h)
      self._h = None

  def __del__(self):
    try:
      self.close()
    except Exception:  # pylint: disable=broad-except
      pass

  def set_pool(self, pool):
    self.pool = pool
    cpool = pool.as_struct()
    check(self.lib.swb_set_pool(self._h, C.byref(cpool)))
    self._rendered = 0

  def sample_pool(self, spec, n_entries, pool_base, pool_len, seed, first_entry=0):
    base = np.ascontiguousarray(pool_base, dtype=np.int32)
    length = np.ascontiguousarray(pool_len, dtype=np.int32)
    check(self.lib.swb_sample_pool(self._h, C.byref(spec), int(n_entries), _ptr(base), _ptr(length),
                                   C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), C.c_uint64(int(first_entry)), None))
    self.pool = None
    self._pool_entries = int(n_entries)
    self._rendered = 0

  def resample_pool(self, seed, first_entry=0):
    check(self.lib.swb_resample_pool(self._h, C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), C.c_uint64(int(first_entry)), None))

  def get_pool(self):
    from spriteworld_amd import lowering
    n = self.pool.n_entries if self.pool is not None else self._pool_entries
    pool = lowering.Pool(n, self.S, self.cfg.n_tasks)
    pool.pool_base = np.zeros(self.N, np.int32)
    pool.pool_len = np.zeros(self.N, np.int32)
    cpool = pool.as_struct()
    check(self.lib.swb_get_pool(self._h, C.byref(cpool)))
    return pool

  def reset_all(self):
    check(self.lib.swb_reset_all(self._h, None))

  def step(self, actions, render=True):
    if self.cfg.action_space == _abi.ACTION_EMBODIED:
      a = np.ascontiguousarray(actions, dtype=np.int32).reshape(self.N, 2)
    else:
      a = np.ascontiguousarray(actions, dtype=np.float32 if self.cfg.action_is_f32 else np.float64).reshape(self.N, 4)
    outs = self._outs(render)
    check(self.lib.swb_step(self._h, _ptr(a), C.byref(outs), None))
    if render:                          # (as engine.Engine: the run lists are trimmed after the third rendering launch)
      self._rendered = getattr(self, '_rendered', 0) + 1
      if self._rendered == 3:
        self.trim()

  def trim(self):
    if not hasattr(self.lib, 'swb_trim_run_lists'):
      return None
    cap = C.c_int32(0)
    check(self.lib.swb_trim_run_lists(self._h, C.byref(cap), None))
    return cap.value

  def render(self):
    check(self.lib.swb_render(self._h, _ptr(self.obs), None))
    return self.obs

  def evaluate(self):
    check(self.lib.swb_evaluate(self._h, _ptr(self.success), None))
    return self.success

  def factors(self):
    out = np.zeros((self.N, self.S, 10), dtype=np.float64)
    check(self.lib.swb_factors(self._h, _ptr(out), None))
    return out

  def state(self):
    st = {
        'x': np.zeros((self.N, self.S)), 'y': np.zeros((self.N, self.S)),
        'n_sprites': np.zeros(self.N, np.int32), 'pool_entry': np.zeros(self.N, np.int32),
        'step_count': np.zeros(self.N, np.int32), 'reset_next': np.zeros(self.N, np.uint8),
        'episode': np.zeros(self.N, np.int32),
    }
    cs = _abi.SwbState(*[a.ctypes.data for a in (st['x'], st['y'], st['n_sprites'], st['pool_entry'],
                                                  st['step_count'], st['reset_next'], st['episode'])])
    check(self.lib.swb_get_state(self._h, C.byref(cs), None))
    return st

  def env_state(self, env):
    out = np.zeros(5, np.int32)
    self.lib.swb_get_env_state.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    check(self.lib.swb_get_env_state(self._h, int(env), _ptr(out), None))
    return dict(zip(('n_sprites', 'pool_entry', 'step_count', 'episode', 'reset_next'), (int(v) for v in out)))

  def sprite_types(self, env, sprite):
    f = C.c_int32(0)
    self.lib.swb_get_sprite_types.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.c_void_p]
    check(self.lib.swb_get_sprite_types(self._h, int(env), int(sprite), C.byref(f), None))
    return bool(f.value & 1), bool(f.value & 2)

  def set_positions(self, x, y):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    check(self.lib.swb_set_positions(self._h, _ptr(x), _ptr(y), None))

  def set_sprite_attr(self, env, sprite, attr, value, delta=None, label=None, cell_label=None):
    d = None if delta is None else C.byref(C.c_double(float(delta)))
    lab = None if label is None else np.ascontiguousarray(label, dtype=np.int8)
    check(self.lib.swb_set_sprite_attr(self._h, int(env), int(sprite), int(attr), float(value), d, _ptr(lab), None))
    if cell_label is not None:
      cells = np.ascontiguousarray(cell_label, dtype=np.int8)
      check(self.lib.swb_set_sprite_cell_labels(self._h, int(env), int(sprite), _ptr(cells), None))

  def get_sprite(self, env, sprite):
    shape, nv = C.c_int32(0), C.c_int32(0)
    angle, scale = C.c_double(0.0), C.c_double(0.0)
    path = np.zeros((_abi.SWB_MAX_SHAPE_VERTS, 2), dtype=np.float64)
    check(self.lib.swb_get_sprite(self._h, int(env), int(sprite), C.byref(shape), C.byref(angle), C.byref(scale),
                                  C.byref(nv), _ptr(path), None))
    return {'shape': shape.value, 'angle': angle.value, 'scale': scale.value, 'path': path[:nv.value].copy()}

  def outputs_host(self):
    return {k: getattr(self, k).copy() for k in ('obs', 'reward', 'discount', 'step_type', 'success', 'error')}

  def variant(self):
    info = _abi.SwbVariantInfo()
    check(self.lib.swb_variant(self._h, C.byref(info)))
    d = {k: getattr(info, k) for k, _ in _abi.SwbVariantInfo._fields_}
    d['cover_kernel'] = 'swb_cover_kernel<%d>' % info.nw
    d['kernel'] = ('swb_resample_kernel<%d>' % info.vs if info.vs else
                   ('none (the cover kernel paints the frame)' if info.paint_in_cover else 'swb_fill_kernel'))
    d['build_id'] = self.lib.swb_build_id().decode()
    return d


class EmuTorchEngine(EmuEngine):
  """EmuEngine whose output buffers are (CPU) torch tensors sharing the numpy arrays' memory: the attribute surface
  `environment.BatchedEnvironment` expects from `engine.Engine` (tests monkeypatch it in)."""

  def __init__(self, cfg, pool, device=0):
    import torch
    EmuEngine.__init__(self, cfg, pool, device)
    self.device = torch.device('cpu')
    self._np = {k: getattr(self, k) for k in ('obs', 'reward', 'discount', 'step_type', 'success', 'error')}
    for k, a in self._np.items():
      setattr(self, k, torch.from_numpy(a))

  def _outs(self, render):
    o = _abi.SwbOutputs()
    o.obs = self._np['obs'].ctypes.data if render else None
    o.reward, o.discount = self._np['reward'].ctypes.data, self._np['discount'].ctypes.data
    o.step_type, o.success, o.error = (self._np['step_type'].ctypes.data, self._np['success'].ctypes.data,
                                        self._np['error'].ctypes.data)
    return o

  def step(self, actions, render=True):
    import torch
    if isinstance(actions, torch.Tensor):
      actions = actions.cpu().numpy()
    EmuEngine.step(self, actions, render=render)

  def render(self):
    check(self.lib.swb_render(self._h, _ptr(self._np['obs']), None))
    return self.obs

  def evaluate(self):
    check(self.lib.swb_evaluate(self._h, _ptr(self._np['success']), None))
    return self.success

  def factors(self):
    import torch
    return torch.from_numpy(EmuEngine.factors(self))

  def outputs_host(self):
    return {k: a.copy() for k, a in self._np.items()}

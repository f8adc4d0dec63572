"""ctypes wrapper of the CPU oracle (oracle/sw_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, `__graft_entry__.smoke()` and
bench.py's `cpu_baseline` leg -- never by the product package.  The oracle is
pinned against the unmodified reference by tests/test_oracle_vs_reference.py and
the golden vectors under tests/golden/ (see sw_oracle.c header).
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np

from spriteworld_amd import _abi
from spriteworld_amd import shapes as _shapes

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, '_build', 'libsw_oracle.so')
_lib = None


def build(force=False):
  """Compiles the oracle with gcc (no-op when up to date).  An exclusive file lock makes concurrent callers (pytest-xdist workers
  right after a source change) wait for one build instead of loading a library another process is still writing."""
  import fcntl
  src = os.path.join(_HERE, 'sw_oracle.c')
  hdr = os.path.join(_HERE, '..', 'include', 'swb.h')

  def fresh():
    return (os.path.exists(_LIB_PATH) and
            os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(src), os.path.getmtime(hdr)))
  if not force and fresh():
    return _LIB_PATH
  os.makedirs(os.path.dirname(_LIB_PATH), exist_ok=True)
  with open(os.path.join(os.path.dirname(_LIB_PATH), '.lock'), 'w') as lock:
    fcntl.flock(lock, fcntl.LOCK_EX)
    if force or not fresh():
      subprocess.check_call(['make', '-C', _HERE, '-B' if force else '-s'], stdout=subprocess.DEVNULL)
  return _LIB_PATH


def lib():
  global _lib
  if _lib is None:
    build()
    _lib = C.CDLL(_LIB_PATH)
    _lib.swo_create.restype = C.c_void_p
    _lib.swo_create.argtypes = [C.c_void_p, C.c_void_p]
    _lib.swo_destroy.argtypes = [C.c_void_p]
    _lib.swo_reset_all.argtypes = [C.c_void_p]
    _lib.swo_step_range.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 7
    _lib.swo_render_range.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    _lib.swo_evaluate_range.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    _lib.swo_get_state.argtypes = [C.c_void_p, C.c_void_p]
    _lib.swo_set_positions.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    _lib.swo_contains_point.argtypes = [C.c_int] + [C.c_double] * 5
    _lib.swo_set_sprite_attr.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p]
    _lib.swo_set_sprite_cell_labels.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    _lib.swo_get_sprite.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 5
    set_shapes()
  return _lib


def set_shapes():
  """(Re)loads the shape table of spriteworld_amd.shapes into the oracle (tests swap a shape for one of 33 .. 64 vertices, which
  the C ABI allows and no built-in shape exercises)."""
  verts, offs = _shapes.packed_table()
  rc = _lib.swo_set_shapes(verts.ctypes.data_as(C.c_void_p), offs.ctypes.data_as(C.c_void_p), C.c_int32(len(offs) - 1))
  assert rc == 0


def _p(a):
  return None if a is None else a.ctypes.data_as(C.c_void_p)


def fill_polygon(width, height, xy, ink=(255, 255, 255), image=None):
  """Pillow ImagingDrawPolygon(fill) on integer vertices; returns u8[H,W,3]."""
  xy = np.ascontiguousarray(xy, dtype=np.int32).reshape(-1, 2)
  if image is None:
    image = np.zeros((height, width, 3), dtype=np.uint8)
  ink = np.asarray(ink, dtype=np.uint8)
  rc = lib().swo_fill_polygon(_p(image), width, height, len(xy), _p(xy), _p(ink))
  assert rc == 0
  return image


def resample(src, out_w, out_h):
  """Pillow Image.resize(LANCZOS) for an RGB u8[Hc,Wc,3] array."""
  src = np.ascontiguousarray(src, dtype=np.uint8)
  hc, wc = src.shape[:2]
  dst = np.zeros((out_h, out_w, 3), dtype=np.uint8)
  lib().swo_resample(_p(src), wc, hc, _p(dst), out_w, out_h)
  return dst


def lanczos_tables(in_size, out_size):
  ks = lib().swo_lanczos_ksize(in_size, out_size)
  bounds = np.zeros((out_size, 2), dtype=np.int32)
  kk = np.zeros((out_size, ks), dtype=np.int32)
  lib().swo_lanczos_coeffs(in_size, out_size, _p(bounds), _p(kk))
  return bounds, kk


def contains_point(shape_index, scale, angle_deg, tx, ty):
  """matplotlib Path.contains_point of the centred sprite path at (tx, ty)."""
  th = math.radians(angle_deg)
  return bool(lib().swo_contains_point(shape_index, float(scale), math.cos(th), math.sin(th),
                                       float(tx), float(ty)))


def vertices(shape_index, scale, angle_deg, px, py):
  th = math.radians(angle_deg)
  out = np.zeros((_abi.SWB_MAX_SHAPE_VERTS, 2), dtype=np.float64)
  lib().swo_vertices.argtypes = [C.c_int] + [C.c_double] * 5 + [C.c_void_p]
  n = lib().swo_vertices(shape_index, float(scale), math.cos(th), math.sin(th), float(px), float(py),
                         _p(out))
  return out[:n].copy()


def davies_bouldin(pos_f32, x, y, label):
  x = np.ascontiguousarray(x, dtype=np.float64)
  y = np.ascontiguousarray(y, dtype=np.float64)
  label = np.ascontiguousarray(label, dtype=np.int8)
  out = C.c_double(0.0)
  lib().swo_davies_bouldin.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
  err = lib().swo_davies_bouldin(int(pos_f32), len(x), _p(x), _p(y), _p(label), C.byref(out))
  return err, out.value


def render_sprites(cfg, x, y, shape, scale, cos_a, sin_a, rgb):
  """One frame (PILRenderer.render) for explicit sprite arrays."""
  n = len(x)
  f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
  x, y, scale, cos_a, sin_a = f(x), f(y), f(scale), f(cos_a), f(sin_a)
  shape = np.ascontiguousarray(shape, dtype=np.int32)
  rgb4 = np.zeros((max(n, 1), 4), dtype=np.uint8)
  if n:
    rgb4[:, :3] = np.asarray(rgb, dtype=np.uint8).reshape(n, 3)
  obs = np.zeros((cfg.image_w, cfg.image_h, 3), dtype=np.uint8)
  lib().swo_render_sprites(C.byref(cfg), n, _p(x), _p(y), _p(shape), _p(scale), _p(cos_a), _p(sin_a),
                           _p(rgb4), _p(obs))
  return obs


class Engine(object):
  """Batched CPU oracle with the same inputs as the HIP engine.

  `cfg` is a spriteworld_amd._abi.SwbConfig, `pool` a
  spriteworld_amd.lowering.Pool (host arrays).
  """

  def __init__(self, cfg, pool):
    self.cfg = cfg
    self.pool = pool  # keep arrays alive
    self._cpool = pool.as_struct()
    self._h = lib().swo_create(C.byref(cfg), C.byref(self._cpool))
    self.N, self.S = cfg.n_envs, cfg.max_sprites
    self.obs_shape = (cfg.image_w, cfg.image_h, 3)

  def __del__(self):
    if getattr(self, '_h', None):
      lib().swo_destroy(self._h)
      self._h = None

  def reset_all(self):
    lib().swo_reset_all(self._h)

  def step(self, actions, render=True, env_range=None):
    cfg = self.cfg
    if cfg.action_space == _abi.ACTION_EMBODIED:
      actions = np.ascontiguousarray(actions, dtype=np.int32).reshape(self.N, 2)
    else:
      adt = np.float32 if cfg.action_is_f32 else np.float64
      actions = np.ascontiguousarray(actions, dtype=adt).reshape(self.N, 4)
    i0, i1 = (0, self.N) if env_range is None else env_range
    out = {
        'obs': np.zeros((self.N,) + self.obs_shape, dtype=np.uint8) if render else None,
        'reward': np.zeros(self.N, dtype=np.float64),
        'discount': np.zeros(self.N, dtype=np.float32),
        'step_type': np.zeros(self.N, dtype=np.uint8),
        'success': np.zeros(self.N, dtype=np.uint8),
        'error': np.zeros(self.N, dtype=np.uint8),
    }
    lib().swo_step_range(self._h, i0, i1, _p(actions), _p(out['obs']), _p(out['reward']),
                         _p(out['discount']), _p(out['step_type']), _p(out['success']),
                         _p(out['error']))
    return out

  def render(self):
    obs = np.zeros((self.N,) + self.obs_shape, dtype=np.uint8)
    lib().swo_render_range(self._h, 0, self.N, _p(obs))
    return obs

  def evaluate(self):
    """environment.py:80-81: task.success() of the current sprites."""
    ok = np.zeros(self.N, dtype=np.uint8)
    lib().swo_evaluate_range(self._h, 0, self.N, _p(ok))
    return ok

  def state(self):
    st = {
        'x': np.zeros((self.N, self.S)), 'y': np.zeros((self.N, self.S)),
        'n_sprites': np.zeros(self.N, np.int32), 'pool_entry': np.zeros(self.N, np.int32),
        'step_count': np.zeros(self.N, np.int32), 'reset_next': np.zeros(self.N, np.uint8),
        'episode': np.zeros(self.N, np.int32),
    }
    cs = _abi.SwbState(*[a.ctypes.data for a in (st['x'], st['y'], st['n_sprites'], st['pool_entry'],
                                                  st['step_count'], st['reset_next'], st['episode'])])
    lib().swo_get_state(self._h, C.byref(cs))
    return st

  def set_positions(self, x, y):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    lib().swo_set_positions(self._h, _p(x), _p(y))

  def set_sprite_attr(self, env, sprite, attr, value, delta=None, label=None, cell_label=None):
    """sprite.py:152-175 setters on a live sprite (attr: _abi.ATTR_SHAPE / ATTR_ANGLE / ATTR_SCALE).  cell_label:
    i8[n_tasks, SWB_MAX_CELLS] for tasks that key on position (swb_task::n_xcuts)."""
    d = None if delta is None else C.byref(C.c_double(float(delta)))
    lab = None if label is None else np.ascontiguousarray(label, dtype=np.int8)
    rc = lib().swo_set_sprite_attr(self._h, int(env), int(sprite), int(attr), float(value), d, _p(lab))
    if rc != 0:
      raise ValueError('swo_set_sprite_attr failed (%d)' % rc)
    if cell_label is not None:
      cells = np.ascontiguousarray(cell_label, dtype=np.int8)
      rc = lib().swo_set_sprite_cell_labels(self._h, int(env), int(sprite), _p(cells))
      if rc != 0:
        raise ValueError('swo_set_sprite_cell_labels failed (%d)' % rc)

  def get_sprite(self, env, sprite):
    """dict(shape=index, angle, scale, path=f64[n,2]): the sprite as the oracle currently sees it."""
    shape, nv = C.c_int32(0), C.c_int32(0)
    angle, scale = C.c_double(0.0), C.c_double(0.0)
    path = np.zeros((_abi.SWB_MAX_SHAPE_VERTS, 2), dtype=np.float64)
    rc = lib().swo_get_sprite(self._h, int(env), int(sprite), C.byref(shape), C.byref(angle), C.byref(scale),
                              C.byref(nv), _p(path))
    if rc != 0:
      raise ValueError('swo_get_sprite failed (%d)' % rc)
    return {'shape': shape.value, 'angle': angle.value, 'scale': scale.value, 'path': path[:nv.value].copy()}

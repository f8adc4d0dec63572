"""Shared helpers for the tests."""
import contextlib
import ctypes
import glob
import os
import zlib

import numpy as np

from spriteworld_amd import _abi
from spriteworld_amd import lowering

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@contextlib.contextmanager
def swapped_shape(name, verts):
  """`name` of the shape table replaced by `verts` for the engines created inside the block (the oracle's table follows)."""
  from oracle import oracle
  from spriteworld_amd import shapes
  oracle.lib()
  old = shapes.SHAPES[name]
  shapes.SHAPES[name] = np.asarray(verts, dtype=np.float64)
  oracle.set_shapes()
  try:
    yield
  finally:
    shapes.SHAPES[name] = old
    oracle.set_shapes()


def golden_cases():
  return sorted(os.path.splitext(os.path.basename(p))[0]
                for p in glob.glob(os.path.join(GOLDEN_DIR, '*.npz')))


def load_golden(name):
  """Returns (SwbConfig, Pool, npz dict) of a fixture written by tests/golden/make_golden.py."""
  z = np.load(os.path.join(GOLDEN_DIR, name + '.npz'))
  raw = z['cfg_bytes'].tobytes()
  assert len(raw) == ctypes.sizeof(_abi.SwbConfig), 'SwbConfig layout changed: regenerate golden'
  cfg = _abi.SwbConfig.from_buffer_copy(raw)
  P = int(z['pool_n_sprites'].shape[0])
  pool = lowering.Pool(P, cfg.max_sprites, cfg.n_tasks)
  for f in lowering.Pool.FIELDS:
    setattr(pool, f, np.ascontiguousarray(z['pool_' + f]))
  if 'pool_cell_label' in z.files:        # tasks whose filters key on position (round 6)
    pool.cell_label = np.ascontiguousarray(z['pool_cell_label'])
  return cfg, pool, z


def bits64(a):
  return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def check_against_golden(engine, cfg, z, get_state, step, what):
  """Steps `engine` through the fixture's actions and compares with the reference's outputs."""
  n_steps = z['actions'].shape[0]
  for t in range(n_steps):
    out = step(z['actions'][t][None])
    st = get_state()
    msg = '%s step %d' % (what, t)
    assert not out['error'][0], msg
    assert out['step_type'][0] == z['ref_step_type'][t], msg
    n = int(z['ref_n_sprites'][t])
    assert st['n_sprites'][0] == n, msg
    assert np.array_equal(bits64(st['x'][0, :n]), bits64(z['ref_x'][t, :n])), msg
    assert np.array_equal(bits64(st['y'][0, :n]), bits64(z['ref_y'][t, :n])), msg
    assert out['success'][0] == z['ref_success'][t], msg
    r, rr = out['reward'][0], z['ref_reward'][t]
    assert (np.isnan(r) and np.isnan(rr)) or bits64(r) == bits64(rr), (msg, r, rr)
    d, dd = out['discount'][0], z['ref_discount'][t]
    assert (np.isnan(d) and np.isnan(dd)) or d == dd, msg
    frame = np.ascontiguousarray(out['obs'][0])
    assert zlib.crc32(frame.tobytes()) == int(z['ref_frame_crc'][t]), msg + ' frame crc'
    if t < z['frames'].shape[0]:
      assert np.array_equal(frame, z['frames'][t]), msg + ' frame'

"""The reference's own outputs (tests/golden/*.npz, generated from the unmodified reference by
tests/golden/make_golden.py) against the CPU oracle and, on a GPU, the HIP engine.

Bar: step types, positions, rewards, discounts, success flags bit-exact; frames +-0."""
import json
import os

import numpy as np
import pytest

from tests import _util


@pytest.mark.parametrize('name', _util.golden_cases())
def test_oracle_reproduces_reference(name):
  from oracle import oracle
  cfg, pool, z = _util.load_golden(name)
  eng = oracle.Engine(cfg, pool)
  _util.check_against_golden(eng, cfg, z, eng.state, lambda a: eng.step(a), 'oracle/' + name)


@pytest.mark.gpu
@pytest.mark.parametrize('name', _util.golden_cases())
def test_hip_engine_reproduces_reference(name):
  from spriteworld_amd import engine
  cfg, pool, z = _util.load_golden(name)
  eng = engine.Engine(cfg, pool)

  def step(a):
    eng.step(a)
    return eng.outputs_host()

  _util.check_against_golden(eng, cfg, z, eng.state, step, 'hip/' + name)
  eng.close()


def test_shape_tables_match_reference_values():
  from spriteworld_amd import shapes
  with open(os.path.join(_util.GOLDEN_DIR, 'shapes.json')) as f:
    ref = json.load(f)
  assert set(ref) == set(shapes.SHAPES)
  for name, rows in ref.items():
    want = np.array([[float.fromhex(a), float.fromhex(b)] for a, b in rows])
    got = shapes.SHAPES[name]
    assert got.shape == want.shape
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), name
  verts, offs = shapes.packed_table()
  assert list(np.diff(offs)) == [3, 4, 5, 6, 8, 30, 8, 10, 12, 12, 15, 18]

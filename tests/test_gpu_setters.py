"""GPU parity of the sprite attribute setters (reference sprite.py:152-175): the HIP
engine through the C ABI (swb_set_sprite_attr / swb_get_sprite, the OV builds of the step kernel) against the CPU
oracle, whose setters tests/test_sprite_setters.py pins against the unmodified reference.  Bit-exact state and
rewards, frames +-0.  The scenarios live in tests/_setter_cases.py."""
import pytest

from spriteworld_amd import _lib
from tests import _setter_cases as cases

pytestmark = pytest.mark.gpu


def _engine(cfg, pool):
  from spriteworld_amd import engine
  return engine.Engine(cfg, pool)


def test_setters_goal_s5_aa5():
  cases.run_parity(_engine, 'goal_s5', 128, 16, 5)


def test_setters_cluster_s5_aa1():
  cases.run_parity(_engine, 'cluster_s5', 128, 12, 1)


def test_setters_embodied_s12_128x128():
  cases.run_parity(_engine, 'embodied_s12', 32, 10, 5)


def test_setters_ragged_sprite_counts():
  cases.run_parity(_engine, 'ragged_s16', 64, 10, 5)


@pytest.mark.parametrize('geom,aa', [('256x64', 2), ('96x48', 3), ('32x32', 8), ('64x256', 1)])
def test_setters_image_geometries(geom, aa):
  """The OV build of every kernel variant (canvas words x output columns x rows in flight)."""
  cases.run_parity(_engine, 'geom_' + geom, 24, 5, aa)


def test_factors_observation_and_reset_semantics():
  cases.factors_and_reset_case(_engine, _lib.SwbError)


def test_live_sprite_handles_follow_the_reference_setters():
  cases.live_sprite_case()


def test_setter_under_a_filter_that_also_keys_on_position():
  cases.setter_under_a_position_filter_case()

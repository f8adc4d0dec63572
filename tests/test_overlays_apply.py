"""The experiment overlays of tools/overlays (profiling cuts, wave timelines, queued kernel variants) are text edits of the
shipped kernel sources anchored on exact strings: every one of them must still apply to the sources as they are -- an overlay
whose anchor went stale fails here, on the CPU, and not in a GPU session."""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OVERLAYS = [['phase_cut:3'], ['phase_cut:4'], ['phase_cut:5'], ['phase_cut:1'], ['trace'], ['trace_p2'],
            ['nofile'], ['pow_memory'], ['bisect_r06:cells,grow'], ['trace_p1b'], ['define:SWB_RS_WAVES_PER_BLOCK=8'], ['cover_snake']]


def _overlay_build():
  spec = importlib.util.spec_from_file_location('overlay_build', os.path.join(ROOT, 'tools', 'overlay_build.py'))
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  return mod


@pytest.mark.parametrize('overlays', OVERLAYS, ids=['+'.join(o) for o in OVERLAYS])
def test_overlay_applies_to_the_shipped_sources(overlays, tmp_path, monkeypatch):
  ob = _overlay_build()
  name = 'pytest_' + '_'.join(o.replace(':', '') for o in overlays)
  work, csrc = ob.make_copy(name, overlays)
  try:
    with open(os.path.join(csrc, 'swb_kernels.hip.inc')) as f:
      edited = f.read()
    with open(os.path.join(ROOT, 'spriteworld_amd', 'csrc', 'swb_kernels.hip.inc')) as f:
      shipped = f.read()
    assert edited != shipped
  finally:
    import shutil
    shutil.rmtree(work, ignore_errors=True)


def test_every_overlay_file_is_listed():
  names = {os.path.splitext(f)[0] for f in os.listdir(os.path.join(ROOT, 'tools', 'overlays')) if f.endswith('.py')}
  listed = {o.split(':')[0] for group in OVERLAYS for o in group}
  assert names == listed, names ^ listed


def test_event_counters_of_the_emulated_build_still_anchor():
  """tools/emu_stats.py (the exact event counts behind the resample cost model, tools/assemble_evidence.py) instruments the kernel
  source at anchor lines (tests/emu/build_emu.py): they must all still be there."""
  sys.path.insert(0, os.path.join(ROOT, 'tests', 'emu'))
  import build_emu
  with open(os.path.join(ROOT, 'spriteworld_amd', 'csrc', 'swb_kernels.hip.inc')) as f:
    text = f.read()
  assert 'emu_count(' in build_emu._instrument(text)

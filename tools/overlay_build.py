#!/usr/bin/env python
"""Experiment builds of the engine: a copy of spriteworld_amd/csrc with named overlays applied, compiled for gfx950 into
spriteworld_amd/csrc/exp_NAME.so (git-ignored; travels to the GPU box; load it with SWB_LIBRARY=...).

  python tools/overlay_build.py NAME [overlay[:ARG] ...] [-DMACRO=VALUE ...] [--emu]

An overlay is tools/overlays/<overlay>.py with `apply(files, arg)`: `files` maps a source file name to its text and is edited
in place through `replace_once` (every anchor must match exactly once, so an overlay that no longer fits the sources fails
loudly).  Timing experiments, phase cuts and wave timelines live there -- never in the shipped sources.  --emu also runs
the emulated parity suite on the copy (overlays that must not change results)."""
import importlib.util
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spriteworld_amd import build  # noqa: E402


def replace_once(files, name, old, new, count=1):
  assert files[name].count(old) == count, '%s: anchor matches %d times (want %d): %r' % (name, files[name].count(old), count, old[:80])
  files[name] = files[name].replace(old, new)


def make_copy(name, overlays):
  work = os.path.join('/tmp/swb_overlay', name)
  shutil.rmtree(work, ignore_errors=True)
  csrc = os.path.join(work, 'csrc')
  shutil.copytree(build.CSRC, csrc, ignore=shutil.ignore_patterns('*.so', '*.o', '*.hash'))
  files = {}
  for s in build.SOURCES:
    with open(os.path.join(csrc, s)) as f:
      files[s] = f.read()
  files['swb_kernels.hip.inc'] = files['swb_kernels.hip.inc'].replace(
      '#include "../../include/swb.h"', '#include "%s"' % os.path.join(ROOT, 'include', 'swb.h'))
  for ov in overlays:
    mod_name, _, arg = ov.partition(':')
    spec = importlib.util.spec_from_file_location(mod_name, os.path.join(ROOT, 'tools', 'overlays', mod_name + '.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.apply(files, arg, replace_once)
  for s, text in files.items():
    with open(os.path.join(csrc, s), 'w') as f:
      f.write(text)
  return work, csrc


def compile_copy(name, work, csrc, extra_flags=()):
  objs, procs = [], []
  for unit, extra in build.UNITS:
    obj = os.path.join(work, unit.replace('.hip', '.o'))
    procs.append(subprocess.Popen(['/opt/rocm/bin/hipcc'] + build.COMMON + list(extra) + list(extra_flags) +
                                  ['-DSWB_BUILD_ID="exp_%s"' % name, '-c', '-o', obj, os.path.join(csrc, unit)]))
    objs.append(obj)
  for proc in procs:
    if proc.wait() != 0:
      sys.exit('hipcc failed')
  out = os.path.join(build.CSRC, 'exp_%s.so' % name)
  subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs)
  return out


def main():
  args = [a for a in sys.argv[1:] if not a.startswith('--') and not a.startswith('-D')]
  defines = [a for a in sys.argv[1:] if a.startswith('-D')]
  name, overlays = args[0], args[1:]
  work, csrc = make_copy(name, overlays)
  if '--emu' in sys.argv:
    env = dict(os.environ, SWB_EMU_CSRC=csrc)
    rc = subprocess.call([sys.executable, '-m', 'pytest', 'tests/test_emulated_kernel.py', 'tests/test_reference_kats.py', '-q', '-x',
                          '-k', 'emu', '-p', 'no:cacheprovider'], cwd=ROOT, env=env)
    if rc != 0:
      sys.exit('emulated parity FAILED')
  print('built', compile_copy(name, work, csrc, defines))


if __name__ == '__main__':
  main()

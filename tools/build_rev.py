#!/usr/bin/env python
"""A/B builds: the engine as of a git revision, compiled with the shipped flags into spriteworld_amd/csrc/exp_NAME.so
(git-ignored; travels to the GPU box; load it with SWB_LIBRARY=...).  An older revision may lack exports newer host code
binds: such a build serves tools/quick_bench.py (swb_step and the timing calls), not the test suite.

  python tools/build_rev.py NAME REV        e.g.  python tools/build_rev.py r4final 14e4f12"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spriteworld_amd import build  # noqa: E402


def main():
  name, rev = sys.argv[1], sys.argv[2]
  work = os.path.join('/tmp/swb_rev', name)
  shutil.rmtree(work, ignore_errors=True)
  os.makedirs(work)
  tar = subprocess.Popen(['git', 'archive', rev, 'spriteworld_amd/csrc', 'include'], cwd=ROOT, stdout=subprocess.PIPE)
  subprocess.check_call(['tar', '-x', '-C', work], stdin=tar.stdout)
  if tar.wait() != 0:
    sys.exit('git archive failed')
  csrc = os.path.join(work, 'spriteworld_amd', 'csrc')
  objs, procs = [], []
  for unit, extra in build.UNITS:
    obj = os.path.join(work, unit.replace('.hip', '.o'))
    procs.append(subprocess.Popen(['/opt/rocm/bin/hipcc'] + build.COMMON + list(extra) +
                                  ['-DSWB_BUILD_ID="rev_%s"' % name, '-c', '-o', obj, os.path.join(csrc, unit)], cwd=csrc))
    objs.append(obj)
  for proc in procs:
    if proc.wait() != 0:
      sys.exit('hipcc failed')
  out = os.path.join(build.CSRC, 'exp_%s.so' % name)
  subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs)
  print('built', out)


if __name__ == '__main__':
  main()

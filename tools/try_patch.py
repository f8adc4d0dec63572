#!/usr/bin/env python
"""Takes kernel patches through everything that can be said about them WITHOUT a GPU, then leaves a library for the GPU.

  python tools/try_patch.py NAME tools/next_round/a.patch [b.patch ...]

1. copies spriteworld_amd/csrc to /tmp/swb_next/NAME/csrc and applies the patches (`patch -p3`: paths as in the repo);
2. builds the host emulation of that copy (tests/emu) and runs the emulated parity suite against it
   (tests/test_emulated_kernel.py + the reference KATs with kind='emu': oracle, golden fixtures, setters, sampler);
3. compiles the copy with hipcc for gfx950 exactly like spriteworld_amd/build.py into
   spriteworld_amd/csrc/exp_NAME.so (git-ignored; travels to the GPU box, load it with SWB_LIBRARY=... for an A/B of
   bench.py and `pytest -m gpu`) and prints the ISA resources of every variant next to those of the shipped sources.
Nothing under spriteworld_amd/csrc is modified: adopting a patch is a separate, deliberate `patch -p1` + GPU run."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from spriteworld_amd import build  # noqa: E402


def main():
  name, patches = sys.argv[1], [os.path.abspath(p) for p in sys.argv[2:]]
  work = os.path.join('/tmp/swb_next', name)
  shutil.rmtree(work, ignore_errors=True)
  csrc = os.path.join(work, 'csrc')
  shutil.copytree(build.CSRC, csrc, ignore=shutil.ignore_patterns('*.so', '*.o', '*.hash'))
  text = open(os.path.join(csrc, 'swb_kernels.hip.inc')).read()
  open(os.path.join(csrc, 'swb_kernels.hip.inc'), 'w').write(
      text.replace('#include "../../include/swb.h"', '#include "%s"' % os.path.join(ROOT, 'include', 'swb.h')))
  for p in patches:
    subprocess.check_call(['patch', '-p3', '-d', csrc, '-i', p])
  env = dict(os.environ, SWB_EMU_CSRC=csrc)
  print('== emulated parity suite on the patched sources', flush=True)
  rc = subprocess.call([sys.executable, '-m', 'pytest', 'tests/test_emulated_kernel.py', 'tests/test_reference_kats.py', '-q', '-x',
                        '-k', 'emu', '-p', 'no:cacheprovider'], cwd=ROOT, env=env)
  if rc != 0:
    sys.exit('emulated parity FAILED: not building the GPU library')
  print('== hipcc build (gfx950)', flush=True)
  objs, procs = [], []
  for unit, extra in build.UNITS:
    obj = os.path.join(work, unit.replace('.hip', '.o'))
    procs.append(subprocess.Popen(['hipcc'] + build.COMMON + extra + ['-DSWB_BUILD_ID="exp_%s"' % name, '-c', '-o', obj,
                                                                      os.path.join(csrc, unit)]))
    objs.append(obj)
  for proc in procs:
    if proc.wait() != 0:
      sys.exit('hipcc failed')
  out = os.path.join(build.CSRC, 'exp_%s.so' % name)
  subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs)
  print('built', out)
  print('== ISA resources: patched', flush=True)
  subprocess.check_call([sys.executable, os.path.join(ROOT, 'tools', 'isa_resources.py'), csrc])


if __name__ == '__main__':
  main()

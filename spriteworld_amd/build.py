"""Builds the HIP engine in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
import hashlib
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
SOURCES = ('swb.hip', 'swb_kernels.hip.inc', 'swb_pow.hip.inc', 'swb_pow_tables.inc', 'swb_sampler.hip.inc')
# -amdgpu-sched-strategy=iterative-ilp: the ILP-first list scheduler measured 3 % faster than the
# default on the step kernel (tools/exp_libs.sh), same results bit for bit.
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fno-fast-math',
         '-mllvm', '-amdgpu-sched-strategy=iterative-ilp', '-shared', '-fPIC']


def source_hash():
  """sha256 over the sources, the public header and the flags: the identity of a build."""
  h = hashlib.sha256(' '.join(FLAGS).encode())
  for path in [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(_HERE, '..', 'include', 'swb.h')]:
    with open(path, 'rb') as f:
      h.update(f.read())
  return h.hexdigest()


def build(force=False, verbose=False):
  """Compiles csrc/libswb.so.  Without `force` a library built from exactly these sources and flags
  (content hash recorded beside it in libswb.so.hash) is reused."""
  out = os.path.join(CSRC, 'libswb.so')
  stamp = out + '.hash'
  want = source_hash()
  if not force and os.path.exists(out) and os.path.exists(stamp):
    with open(stamp) as f:
      if f.read().strip() == want:
        return out
  hipcc = os.environ.get('HIPCC', 'hipcc')
  if not any(os.access(os.path.join(d, hipcc), os.X_OK) for d in os.environ['PATH'].split(':')):
    hipcc = '/opt/rocm/bin/hipcc'
  cmd = [hipcc] + FLAGS + ['-DSWB_BUILD_ID="%s"' % want[:16], '-o', out, os.path.join(CSRC, 'swb.hip')]
  if verbose:
    print(' '.join(cmd))
  subprocess.check_call(cmd, cwd=CSRC)
  with open(stamp, 'w') as f:
    f.write(want + '\n')
  return out


def built_hash():
  """Content hash the shipped libswb.so was built from (None when unknown)."""
  stamp = os.path.join(CSRC, 'libswb.so.hash')
  if not os.path.exists(stamp):
    return None
  with open(stamp) as f:
    return f.read().strip()

"""Builds the HIP engine in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
import hashlib
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
SOURCES = ('swb.hip', 'swb_wide.hip', 'swb_kernels.hip.inc', 'swb_pow.hip.inc', 'swb_pow_tables.inc', 'swb_sampler.hip.inc')
COMMON = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fno-fast-math', '-fPIC']
# Two translation units, two instruction schedulers (same results bit for bit; measured on MI355X, round 2):
#   swb.hip       host code + the kernels of images up to 64 columns: LLVM's default strategy -- as fast as the
#                 ILP-first one at 4 waves per SIMD (0.243 vs 0.247 ms) with 11 instead of 69 spilled VGPRs;
#   swb_wide.hip  the kernels of wider images (2-3 waves per SIMD): -amdgpu-sched-strategy=iterative-ilp,
#                 5 % faster there (2.31 vs 2.43 ms on 12 sprites, 128x128).
UNITS = (('swb.hip', []), ('swb_wide.hip', ['-mllvm', '-amdgpu-sched-strategy=iterative-ilp']))
FLAGS = COMMON + ['units=' + ';'.join('%s:%s' % (u, ' '.join(f)) for u, f in UNITS)]      # hashed with the sources


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
  objs, procs = [], []
  for unit, extra in UNITS:                       # the two units compile in parallel
    obj = os.path.join(CSRC, unit.replace('.hip', '.o'))
    cmd = [hipcc] + COMMON + extra + ['-DSWB_BUILD_ID="%s"' % want[:16], '-c', '-o', obj, os.path.join(CSRC, unit)]
    if verbose:
      print(' '.join(cmd))
    procs.append((cmd, subprocess.Popen(cmd, cwd=CSRC)))
    objs.append(obj)
  for cmd, proc in procs:
    if proc.wait() != 0:
      raise subprocess.CalledProcessError(proc.returncode, cmd)
  cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs
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

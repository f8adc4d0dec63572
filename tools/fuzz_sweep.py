#!/usr/bin/env python
"""Randomised parity sweep on the GPU box: the seeded random configurations of tests/test_gpu_parity.py
(geometry x anti-aliasing x sprite counts x shapes x task x action space x dtype) for a range of seeds beyond
the 24 the test suite runs, HIP engine vs oracle, everything bit-exact / frames +-0.
usage: python tools/fuzz_sweep.py FIRST LAST [OUT.txt] [--big-polygons]
--big-polygons: every third seed runs with the circle of the shape table swapped for a regular polygon of 33 .. 64 vertices
(33 + seed % 32) on both sides -- shapes the C ABI allows and no built-in shape exercises (one lane per edge in the scatter)."""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import test_gpu_parity as T  # noqa: E402


def main():
  big = '--big-polygons' in sys.argv
  argv = [a for a in sys.argv if not a.startswith('--')]
  first, last = int(argv[1]), int(argv[2])
  out = open(argv[3], 'w') if len(argv) > 3 else sys.stdout
  bad = []
  n_big = 0
  for seed in range(first, last):
    try:
      if big and seed % 3 == 0:
        from spriteworld_amd import shapes
        from tests import _util
        n_big += 1
        with _util.swapped_shape('circle', shapes.polygon(33 + seed % 32)):
          T._run('fuzz_%d' % seed, 64, 10, 5, seed=seed)
      else:
        T._run('fuzz_%d' % seed, 64, 10, 5, seed=seed)
    except Exception:  # pylint: disable=broad-except
      bad.append(seed)
      out.write('seed %d FAILED\n%s\n' % (seed, traceback.format_exc()[-1500:]))
      out.flush()
    if (seed + 1 - first) % 50 == 0:          # (progress: a sweep cut by a timeout still says how far it got)
      out.write('... seeds [%d, %d): %d failed so far\n' % (first, seed + 1, len(bad)))
      out.flush()
  out.write('fuzz seeds [%d, %d): %d passed, %d failed %s%s\n' % (first, last, last - first - len(bad), len(bad), bad,
                                                                  ' (%d of them with a 33 .. 64-gon in place of the circle)' % n_big if big else ''))
  out.flush()


if __name__ == '__main__':
  main()

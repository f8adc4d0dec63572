#!/usr/bin/env python
"""Randomised parity sweep on the GPU box: the seeded random configurations of tests/test_gpu_parity.py
(geometry x anti-aliasing x sprite counts x shapes x task x action space x dtype) for a range of seeds beyond
the 24 the test suite runs, HIP engine vs oracle, everything bit-exact / frames +-0.
usage: python tools/fuzz_sweep.py FIRST LAST [OUT.txt]"""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import test_gpu_parity as T  # noqa: E402


def main():
  first, last = int(sys.argv[1]), int(sys.argv[2])
  out = open(sys.argv[3], 'w') if len(sys.argv) > 3 else sys.stdout
  bad = []
  for seed in range(first, last):
    try:
      T._run('fuzz_%d' % seed, 64, 10, 5, seed=seed)
    except Exception:  # pylint: disable=broad-except
      bad.append(seed)
      out.write('seed %d FAILED\n%s\n' % (seed, traceback.format_exc()[-1500:]))
      out.flush()
    if (seed + 1 - first) % 50 == 0:          # (progress: a sweep cut by a timeout still says how far it got)
      out.write('... seeds [%d, %d): %d failed so far\n' % (first, seed + 1, len(bad)))
      out.flush()
  out.write('fuzz seeds [%d, %d): %d passed, %d failed %s\n' % (first, last, last - first - len(bad), len(bad), bad))
  out.flush()


if __name__ == '__main__':
  main()

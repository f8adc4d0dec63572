import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)


# Small batches (a round of waves or less even with a wave per band of output rows) run the TEAM build of the cover kernel, and
# nearly every test batch is small.  The suite keeps its tests on the build the large batches -- the benchmark's -- run, as in
# rounds 1 - 5; the TEAM build has tests of its own (`team` in their names), which lift this.
os.environ.setdefault('SWB_NO_TEAM', '1')


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


def _has_gpu():
  try:
    import torch
    return torch.cuda.is_available()
  except Exception:  # pylint: disable=broad-except
    return False


def pytest_collection_modifyitems(config, items):
  if _has_gpu():
    return
  skip = pytest.mark.skip(reason='no GPU visible')
  for item in items:
    if 'gpu' in item.keywords:
      item.add_marker(skip)

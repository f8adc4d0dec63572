"""The C-ABI library loads on a CPU-only host and exports every symbol include/swb.h declares;
struct layouts of the ctypes mirror equal the C ones (no compute calls without a GPU)."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

from spriteworld_amd import _abi, _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'swb.h')


def _declared_functions():
  text = open(HEADER).read()
  return sorted(set(re.findall(r'^\s*(?:const char\*|int)\s+(swb_\w+)\s*\(', text, flags=re.M)))


def test_header_and_export_list_agree():
  assert _declared_functions() == sorted(_lib.EXPORTS)


def test_library_builds_loads_and_exports_all_symbols():
  build.build()
  lib = ctypes.CDLL(_lib.LIB_PATH)
  for name in _declared_functions():
    assert hasattr(lib, name), name
  lib.swb_version.restype = ctypes.c_int
  assert lib.swb_version() >= 1


def test_struct_layouts_match_the_header(tmp_path):
  src = tmp_path / 'sizes.c'
  src.write_text('''
#include <stdio.h>
#include <stddef.h>
#include "swb.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(swb_task), sizeof(swb_config), sizeof(swb_pool),
         sizeof(swb_outputs), sizeof(swb_state), sizeof(swb_factor), sizeof(swb_sprite_group), sizeof(swb_sampler));
  printf("%zu %zu %zu\\n", sizeof(swb_holdout), sizeof(swb_alternative), sizeof(swb_variant_info));
  printf("%zu %zu %zu %zu\\n", offsetof(swb_config, action_scale), offsetof(swb_config, n_tasks),
         offsetof(swb_config, meta_terminate_bonus), offsetof(swb_config, tasks));
  return 0;
}''')
  exe = tmp_path / 'sizes'
  subprocess.check_call(['gcc', '-I', os.path.join(ROOT, 'include'), '-o', str(exe), str(src)])
  out = subprocess.check_output([str(exe)]).decode().split()
  got = [ctypes.sizeof(c) for c in (_abi.SwbTask, _abi.SwbConfig, _abi.SwbPool, _abi.SwbOutputs, _abi.SwbState,
                                     _abi.SwbFactor, _abi.SwbSpriteGroup, _abi.SwbSampler, _abi.SwbHoldout,
                                     _abi.SwbAlternative, _abi.SwbVariantInfo)]
  got += [getattr(_abi.SwbConfig, f).offset for f in ('action_scale', 'n_tasks', 'meta_terminate_bonus', 'tasks')]
  assert [int(v) for v in out] == got


def test_engine_fails_loudly_without_gpu():
  import torch
  if torch.cuda.is_available():
    pytest.skip('GPU present')
  from spriteworld_amd import engine, workloads
  cfg, pool, _ = workloads.build('goal_s5', 4, 1)
  with pytest.raises(_lib.SwbError):
    engine.Engine(cfg, pool)


def test_product_never_imports_the_oracle():
  """The oracle is test infrastructure: nothing under spriteworld_amd/ may import, include or load it."""
  pkg = os.path.join(ROOT, 'spriteworld_amd')
  for dirpath, _, files in os.walk(pkg):
    for f in files:
      path = os.path.join(dirpath, f)
      if f.endswith('.py'):
        for line in open(path):
          assert not re.match(r'\s*(from|import)\s+oracle\b', line), (f, line)
          assert 'libsw_oracle' not in line and 'oracle/_build' not in line, (f, line)
      elif f.endswith(('.hip', '.inc', '.h')):
        for line in open(path):
          assert not (line.lstrip().startswith('#include') and 'oracle' in line), (f, line)


def test_product_package_never_reaches_for_the_checkers():
  """The oracle (oracle/) and the host emulation of the kernels (tests/emu) are test infrastructure: nothing under
  spriteworld_amd/ imports, loads or names them (comments in docstrings aside), so no product path can route through
  a CPU implementation -- without libswb.so and a GPU the engine raises (test_engine_fails_loudly_without_gpu)."""
  import ast
  pkg = os.path.join(ROOT, 'spriteworld_amd')
  for dirpath, _, files in os.walk(pkg):
    for name in files:
      if not name.endswith('.py'):
        continue
      path = os.path.join(dirpath, name)
      tree = ast.parse(open(path).read())
      for node in ast.walk(tree):
        mods = []
        if isinstance(node, ast.Import):
          mods = [a.name for a in node.names]
        elif isinstance(node, ast.ImportFrom):
          mods = [node.module or '']
        for m in mods:
          assert not (m == 'oracle' or m.startswith('oracle.') or m == 'tests' or m.startswith('tests.')), (path, m)
        if isinstance(node, ast.Constant) and isinstance(node.value, str) and node is not getattr(tree.body[0], 'value', None):
          assert 'libswb_emu' not in node.value and 'libsw_oracle' not in node.value, (path, node.value[:60])
  for name in os.listdir(os.path.join(pkg, 'csrc')):
    if name.endswith(('.hip', '.inc')):
      text = open(os.path.join(pkg, 'csrc', name)).read()
      assert '#include "../../oracle' not in text and 'tests/emu/' not in text.replace('tests/emu found it', ''), name

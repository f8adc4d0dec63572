#!/usr/bin/env python
"""Builds `oracle/_ref/`: the UNMODIFIED reference as a compiled artefact (TEST INFRASTRUCTURE ONLY).

`/root/reference` exists only in the build container; the GPU box gets this repository's tree, including
git-ignored build outputs (`oracle/_ref/` is in .gitignore, not in .gpurunignore).  This recipe byte-compiles the
reference's Python modules FROM THE SOURCES WHERE THEY LIE into sourceless `.pyc` files
    /root/reference/spriteworld/**/*.py   ->  oracle/_ref/spriteworld/**/*.pyc
    /root/reference/example_run_loop.py   ->  oracle/_ref/example_run_loop.pyc
exactly as a C reference would be compiled into `oracle/_ref/*.so`: no reference source is copied into the
repository, nothing is patched, and CPython imports a `name.pyc` that sits where `name.py` would
(importlib's SourcelessFileLoader) -- the same interpreter (3.10) runs here and on the GPU box.

Callers: `__graft_entry__.build()` (when /root/reference is present), `oracle/ref_harness.py` (which falls back to
`oracle/_ref` when /root/reference is absent).  With it the reference runs ON THE GPU NODE: timed as bench.py's
`cpu_baseline` (kind "reference") and stepped beside the HIP engine by the `-m gpu` tests.

usage: python oracle/stage_ref.py [REFERENCE_ROOT]
"""
import hashlib
import json
import os
import py_compile
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, '_ref')
SKIP_DIRS = {'tests', '__pycache__', '.git'}


def _sources(root):
  """(absolute source path, path relative to the staging root) of every module to compile."""
  out = []
  pkg = os.path.join(root, 'spriteworld')
  for d, dirs, files in os.walk(pkg):
    dirs[:] = sorted(x for x in dirs if x not in SKIP_DIRS)
    for f in sorted(files):
      if f.endswith('.py'):
        src = os.path.join(d, f)
        out.append((src, os.path.relpath(src, root)))
  loop = os.path.join(root, 'example_run_loop.py')
  if os.path.exists(loop):
    out.append((loop, 'example_run_loop.py'))
  return out


def stage(root='/root/reference', verbose=False):
  """Compiles the reference under `root` into oracle/_ref; returns the manifest (None when `root` is absent)."""
  if not os.path.isdir(os.path.join(root, 'spriteworld')):
    return None
  srcs = _sources(root)
  digest = hashlib.sha256()
  for src, rel in srcs:
    with open(src, 'rb') as f:
      digest.update(rel.encode() + b'\0' + f.read())
  manifest_path = os.path.join(OUT, 'MANIFEST.json')
  want = {'source_sha256': digest.hexdigest(), 'python': '%d.%d' % sys.version_info[:2], 'modules': [rel for _, rel in srcs],
          'from': root, 'what': 'sourceless bytecode of the unmodified reference (py_compile), see oracle/stage_ref.py'}
  if os.path.exists(manifest_path):
    with open(manifest_path) as f:
      have = json.load(f)
    if all(have.get(k) == want[k] for k in ('source_sha256', 'python', 'modules')) and \
        all(os.path.exists(os.path.join(OUT, rel + 'c')) for rel in want['modules']):
      return have
  if os.path.isdir(OUT):
    shutil.rmtree(OUT)
  for src, rel in srcs:
    dst = os.path.join(OUT, rel + 'c')                 # name.py -> name.pyc beside where name.py would be
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    # dfile: the path shown in tracebacks; unchecked-hash pycs never look for their source
    py_compile.compile(src, cfile=dst, dfile=os.path.join('<reference>', rel), doraise=True,
                       invalidation_mode=py_compile.PycInvalidationMode.UNCHECKED_HASH)
    if verbose:
      print('compiled', rel)
  with open(manifest_path, 'w') as f:
    json.dump(want, f, indent=1)
  return want


if __name__ == '__main__':
  m = stage(sys.argv[1] if len(sys.argv) > 1 else '/root/reference', verbose=True)
  print('no reference tree' if m is None else 'staged %d modules into %s' % (len(m['modules']), OUT))

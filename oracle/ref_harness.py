"""Loader for the unmodified reference (TEST INFRASTRUCTURE ONLY).

Imports `/root/reference/spriteworld` read-only under the three shims SURVEY.md
§8c / Appendix B pins:
  1. `oracle/compat/dm_env`           -- dm_env is not installed in this image (likewise `oracle/compat/absl`,
                                         which only the reference's example_run_loop.py imports);
  2. `PIL.Image.ANTIALIAS = LANCZOS`  -- alias removed in Pillow 10, used at
                                         spriteworld/renderers/pil_renderer.py:84;
  3. `np.cast[dtype]` shim            -- removed in numpy 2, used at
                                         spriteworld/factor_distributions.py:102.
`/root/reference` exists only in the build container.  Where it is absent (the GPU box) the loader falls
back to `oracle/_ref/`: the same unmodified modules as sourceless bytecode, compiled from the sources where
they lie by `oracle/stage_ref.py` (run by `__graft_entry__.build()`; a git-ignored build output that travels
with the tree like the built `.so` files).  `reference_available()` is False only when neither exists.
Nothing in the product package (`spriteworld_amd/`) imports this module.
"""
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
STAGED_ROOT = os.path.join(_HERE, '_ref')
_COMPAT = os.path.join(_HERE, 'compat')


def _usable(root):
  """A reference tree: the package directory exists and, for the staged bytecode, was compiled by this interpreter version
  (a .pyc of another version cannot be imported: the tree then counts as absent and its users skip)."""
  if not os.path.isdir(os.path.join(root, 'spriteworld')):
    return False
  manifest = os.path.join(root, 'MANIFEST.json')
  if os.path.exists(manifest):
    import json
    try:
      with open(manifest) as f:
        return json.load(f).get('python') == '%d.%d' % sys.version_info[:2]
    except (OSError, ValueError):
      return False
  return True


def _pick_root():
  env = os.environ.get('SPRITEWORLD_REFERENCE')
  for root in ([env] if env else []) + ['/root/reference', STAGED_ROOT]:
    if _usable(root):
      return root
  return env or '/root/reference'


REFERENCE_ROOT = _pick_root()


def reference_available():
  return _usable(REFERENCE_ROOT)


def reference_kind():
  """'source' (the tree itself) or 'bytecode' (oracle/_ref, compiled from it by oracle/stage_ref.py)."""
  return 'bytecode' if os.path.abspath(REFERENCE_ROOT) == os.path.abspath(STAGED_ROOT) else 'source'


def third_party_versions():
  """Versions of the libraries whose arithmetic the reference delegates to, as imported here."""
  import importlib
  out = {}
  for name, mod in (('numpy', 'numpy'), ('pillow', 'PIL'), ('matplotlib', 'matplotlib'), ('scikit-learn', 'sklearn')):
    try:
      out[name] = importlib.import_module(mod).__version__
    except Exception as e:  # pylint: disable=broad-except
      out[name] = 'missing: %r' % (e,)
  return out


def load_reference():
  """Returns the reference `spriteworld` package (imported from REFERENCE_ROOT)."""
  if not reference_available():
    raise RuntimeError('reference tree not present at ' + REFERENCE_ROOT)
  import numpy as np
  from PIL import Image
  if not hasattr(Image, 'ANTIALIAS'):
    Image.ANTIALIAS = Image.LANCZOS
  if not hasattr(np, 'cast'):

    class _Cast(object):

      def __getitem__(self, dtype):
        return lambda x: np.asarray(x).astype(dtype)

    np.cast = _Cast()
  # Never write __pycache__ into the reference: this process and every child it spawns.
  sys.dont_write_bytecode = True
  os.environ['PYTHONDONTWRITEBYTECODE'] = '1'
  # Appended, not prepended: the reference tree also holds a top-level `tests` package that must
  # not shadow this repository's.
  try:
    import dm_env  # noqa: F401  (a real dm_env wins if one is ever installed)
    import absl  # noqa: F401    (only the reference's example_run_loop.py needs it)
  except ImportError:
    if _COMPAT not in sys.path:
      sys.path.append(_COMPAT)
  if REFERENCE_ROOT not in sys.path:
    sys.path.append(REFERENCE_ROOT)
  import spriteworld
  from spriteworld import environment, sprite, tasks, action_spaces  # noqa: F401
  from spriteworld import renderers, factor_distributions  # noqa: F401
  from spriteworld import sprite_generators, constants, shapes  # noqa: F401
  return spriteworld

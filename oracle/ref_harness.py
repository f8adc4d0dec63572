"""Loader for the unmodified reference (TEST INFRASTRUCTURE ONLY).

Imports `/root/reference/spriteworld` read-only under the three shims SURVEY.md
§8c / Appendix B pins:
  1. `oracle/compat/dm_env`           -- dm_env is not installed in this image (likewise `oracle/compat/absl`,
                                         which only the reference's example_run_loop.py imports);
  2. `PIL.Image.ANTIALIAS = LANCZOS`  -- alias removed in Pillow 10, used at
                                         spriteworld/renderers/pil_renderer.py:84;
  3. `np.cast[dtype]` shim            -- removed in numpy 2, used at
                                         spriteworld/factor_distributions.py:102.
`/root/reference` exists only in the build container; on the GPU box
`reference_available()` is False and every caller must skip.  Nothing in the
product package (`spriteworld_amd/`) imports this module.
"""
import os
import sys

REFERENCE_ROOT = os.environ.get('SPRITEWORLD_REFERENCE', '/root/reference')
_COMPAT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'compat')


def reference_available():
  return os.path.isdir(os.path.join(REFERENCE_ROOT, 'spriteworld'))


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

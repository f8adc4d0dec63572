"""Loader of the HIP engine (spriteworld_amd/csrc/libswb.so, C ABI of include/swb.h).

There is no CPU fallback: if the library is missing or no gfx950 device is
usable, the engine raises.  Build with `python __graft_entry__.py` (or
`spriteworld_amd.build.build()`).
"""
import ctypes as C
import os

from spriteworld_amd import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# SWB_LIBRARY: an alternative build of the same ABI (experiments, integrators' own install path)
LIB_PATH = os.environ.get('SWB_LIBRARY') or os.path.join(_HERE, 'csrc', 'libswb.so')

# Every symbol include/swb.h declares (tests check the library exports them all).
EXPORTS = (
    'swb_last_error', 'swb_version', 'swb_create', 'swb_destroy', 'swb_upload_shapes',
    'swb_upload_resample', 'swb_set_pool', 'swb_sample_pool', 'swb_resample_pool', 'swb_get_pool', 'swb_reset_all', 'swb_step', 'swb_render', 'swb_evaluate', 'swb_factors',
    'swb_get_state', 'swb_set_positions', 'swb_variant', 'swb_build_id', 'swb_timing_enable', 'swb_step_time_ms',
    'swb_set_sprite_attr', 'swb_get_sprite', 'swb_sprite_path_op', 'swb_kernel_times_ms', 'swb_get_env_state',
    'swb_get_sprite_types', 'swb_trim_run_lists', 'swb_set_sprite_cell_labels',
)

_lib = None


class SwbError(RuntimeError):
  pass


def load():
  """Returns the ctypes handle of libswb.so (raises SwbError when it is not built)."""
  global _lib
  if _lib is not None:
    return _lib
  if not os.path.exists(LIB_PATH):
    raise SwbError('HIP engine not built: %s is missing (run `python __graft_entry__.py`)' %
                   LIB_PATH)
  # PyTorch bundles its own HIP runtime; it must be the first (and only) one loaded in the
  # process, otherwise a second runtime initialised later sees no devices.
  import torch
  if torch.cuda.is_available():
    torch.cuda.init()
  lib = C.CDLL(LIB_PATH)
  lib.swb_last_error.restype = C.c_char_p
  lib.swb_create.argtypes = [C.POINTER(_abi.SwbConfig), C.c_int, C.POINTER(C.c_void_p)]
  lib.swb_destroy.argtypes = [C.c_void_p]
  lib.swb_upload_shapes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
  lib.swb_upload_resample.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                      C.c_void_p]
  lib.swb_set_pool.argtypes = [C.c_void_p, C.POINTER(_abi.SwbPool)]
  lib.swb_sample_pool.argtypes = [C.c_void_p, C.POINTER(_abi.SwbSampler), C.c_int32, C.c_void_p, C.c_void_p,
                                  C.c_uint64, C.c_uint64, C.c_void_p]
  lib.swb_resample_pool.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
  lib.swb_get_pool.argtypes = [C.c_void_p, C.POINTER(_abi.SwbPool)]
  lib.swb_reset_all.argtypes = [C.c_void_p, C.c_void_p]
  lib.swb_step.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(_abi.SwbOutputs), C.c_void_p]
  lib.swb_render.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
  if hasattr(lib, 'swb_evaluate') or not os.environ.get('SWB_LIBRARY'):     # (A/B builds of older revisions lack it)
    lib.swb_evaluate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
  lib.swb_factors.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
  if hasattr(lib, 'swb_trim_run_lists') or not os.environ.get('SWB_LIBRARY'):
    lib.swb_trim_run_lists.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_void_p]
  if hasattr(lib, 'swb_set_sprite_cell_labels') or not os.environ.get('SWB_LIBRARY'):
    lib.swb_set_sprite_cell_labels.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
  lib.swb_get_state.argtypes = [C.c_void_p, C.POINTER(_abi.SwbState), C.c_void_p]
  lib.swb_set_positions.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
  lib.swb_get_env_state.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
  lib.swb_get_sprite_types.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.c_void_p]
  lib.swb_set_sprite_attr.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_void_p, C.c_void_p,
                                      C.c_void_p]
  lib.swb_get_sprite.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p]
  lib.swb_sprite_path_op.argtypes = [C.c_int32, C.c_double, C.c_double, C.c_int32, C.c_void_p, C.c_void_p]
  lib.swb_variant.argtypes = [C.c_void_p, C.POINTER(_abi.SwbVariantInfo)]
  lib.swb_build_id.restype = C.c_char_p
  lib.swb_timing_enable.argtypes = [C.c_void_p, C.c_int32]
  lib.swb_step_time_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
  lib.swb_kernel_times_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]
  _lib = lib
  return lib


def check(rc):
  if rc != 0:
    raise SwbError('swb error %d: %s' % (rc, load().swb_last_error().decode()))

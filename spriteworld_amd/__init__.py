"""spriteworld_amd -- MI355X-native batched Spriteworld step/render engine.

The hot path (Environment.step(): action apply, reward, termination, PIL-exact
rasterisation) runs as hand-written HIP kernels behind a C ABI
(include/swb.h, spriteworld_amd/csrc/); this package is the thin host side that
mirrors the reference's Python surface.  There is no CPU fallback.
"""
from spriteworld_amd import action_spaces  # noqa: F401
from spriteworld_amd import renderers  # noqa: F401
from spriteworld_amd import shapes  # noqa: F401
from spriteworld_amd import sprite  # noqa: F401
from spriteworld_amd import tasks  # noqa: F401

__version__ = '0.1.0'

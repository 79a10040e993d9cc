"""smallvcm_amd -- MI355X-native drop-in for SmallVCM's VertexCM::RunIteration.

The compute path is the HIP library `smallvcm_amd/csrc/libsmallvcm_amd.so`
(C-ABI: include/smallvcm_amd.h).  This package is the Python host mirror of
the reference's renderer interface (src/renderer.hxx:33-70); it holds no
compute of its own and raises if the HIP library is missing.
"""
from . import _abi  # noqa: F401

__all__ = ["_abi"]


def __getattr__(name):
    # lazy: importing the package must not require the built library
    if name in ("VertexCM", "ShardedVertexCM", "cornell_scene", "load_library", "SCENE_CONFIGS"):
        from . import renderer
        return getattr(renderer, name)
    raise AttributeError(name)

"""Scene files (include/smallvcm_amd.h: vcm_scene_load): `.vcmscene` / Wavefront `.obj` + `.mtl` -> SceneDesc2.
The parsing happens in the library (smallvcm_amd/csrc/scene_file.cpp documents the format); this is the ctypes binding.

    scene = load_scene("tests/scenes/bumpy_room.vcmscene", 1024, 1024)
    r = VertexCM(scene, VertexCM.kVcm, 0.003, 0.75)
"""
import ctypes as C

from ._abi import SceneDesc2


class _Handle:
    def __init__(self, L, h):
        self.L, self.h = L, h

    def __del__(self):
        if self.h:
            self.L.vcm_scene_file_free(self.h)
            self.h = None


def load_scene(path, resx, resy):
    """-> SceneDesc2 (the arrays it points to live as long as the returned object)"""
    from .renderer import load_library
    L = load_library(require_gpu=False)
    L.vcm_scene_load.restype = C.c_void_p
    L.vcm_scene_load.argtypes = [C.c_char_p, C.c_int, C.c_int]
    L.vcm_scene_file_desc.restype = C.POINTER(SceneDesc2)
    L.vcm_scene_file_desc.argtypes = [C.c_void_p]
    L.vcm_scene_file_free.argtypes = [C.c_void_p]
    L.vcm_scene_file_free.restype = None
    L.vcm_scene_load_error.restype = C.c_char_p
    h = L.vcm_scene_load(str(path).encode(), int(resx), int(resy))
    if not h:
        raise ValueError("smallvcm_amd: cannot load %s: %s" % (path, L.vcm_scene_load_error().decode()))
    d = SceneDesc2.from_buffer_copy(L.vcm_scene_file_desc(h).contents)   # the struct (pointers into the handle's arrays)
    d._keep = _Handle(L, h)
    return d

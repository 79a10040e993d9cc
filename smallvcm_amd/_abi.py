"""ctypes mirror of include/smallvcm_amd.h (the C-ABI PODs).

Field order and sizes must match the header exactly; tests/test_abi.py checks
sizeof() against the values the C library reports.
"""
import ctypes as C

VCM_MAX_PRIMS = 32
VCM_MAX_MATERIALS = 16
VCM_MAX_LIGHTS = 8
VCM_MERGE_RECORD_FLOATS = 13

PRIM_TRIANGLE, PRIM_SPHERE = 0, 1
LIGHT_AREA, LIGHT_DIRECTIONAL, LIGHT_POINT, LIGHT_BACKGROUND = 0, 1, 2, 3

# VertexCM::AlgorithmType (reference src/vertexcm.hxx:182-204)
ALGO_LIGHT_TRACE, ALGO_PPM, ALGO_BPM, ALGO_BPT, ALGO_VCM = 0, 1, 2, 3, 4
ALGO_PATH_TRACE, ALGO_EYE_LIGHT = 5, 6        # PathTracer (src/pathtracer.hxx), EyeLight (src/eyelight.hxx)
ALGO_BY_NAME = {"lt": ALGO_LIGHT_TRACE, "ppm": ALGO_PPM, "bpm": ALGO_BPM,
                "bpt": ALGO_BPT, "vcm": ALGO_VCM, "pt": ALGO_PATH_TRACE, "el": ALGO_EYE_LIGHT}

f3 = C.c_float * 3


class Prim(C.Structure):
    _fields_ = [("type", C.c_int), ("matID", C.c_int),
                ("p0", f3), ("p1", f3), ("p2", f3), ("n", f3)]


class Material(C.Structure):
    _fields_ = [("diffuse", f3), ("phong", f3), ("phongExp", C.c_float),
                ("mirror", f3), ("ior", C.c_float)]


class Light(C.Structure):
    _fields_ = [("type", C.c_int), ("p0", f3), ("e1", f3), ("e2", f3),
                ("frameX", f3), ("frameY", f3), ("frameZ", f3),
                ("intensity", f3), ("invArea", C.c_float), ("scale", C.c_float)]


class Camera(C.Structure):
    _fields_ = [("position", f3), ("forward", f3), ("resolution", C.c_float * 2),
                ("rasterToWorld", C.c_float * 16), ("worldToRaster", C.c_float * 16),
                ("imagePlaneDist", C.c_float)]


class SceneDesc(C.Structure):
    _fields_ = [("nPrims", C.c_int), ("prims", Prim * VCM_MAX_PRIMS),
                ("nMaterials", C.c_int), ("materials", Material * VCM_MAX_MATERIALS),
                ("mat2light", C.c_int * VCM_MAX_MATERIALS),
                ("nLights", C.c_int), ("lights", Light * VCM_MAX_LIGHTS),
                ("backgroundLight", C.c_int),
                ("sceneCenter", f3), ("sceneRadius", C.c_float),
                ("invSceneRadiusSqr", C.c_float),
                ("camera", Camera)]

    def tobytes(self):
        return bytes(memoryview(self))

    @classmethod
    def frombytes(cls, b):
        if len(b) != C.sizeof(cls):
            raise ValueError("scene desc blob has %d bytes, expected %d" % (len(b), C.sizeof(cls)))
        return cls.from_buffer_copy(b)


class SceneDesc2(C.Structure):
    """vcm_scene_desc2: pointer + count for primitives, materials and lights.  The arrays are owned by the Python
    object that built it (SceneBuilder keeps them in `_keep`); the library copies them at vcm_create2."""
    _fields_ = [("nPrims", C.c_int), ("prims", C.POINTER(Prim)),
                ("nMaterials", C.c_int), ("materials", C.POINTER(Material)), ("mat2light", C.POINTER(C.c_int)),
                ("nLights", C.c_int), ("lights", C.POINTER(Light)),
                ("backgroundLight", C.c_int),
                ("sceneCenter", f3), ("sceneRadius", C.c_float), ("invSceneRadiusSqr", C.c_float),
                ("camera", Camera)]


class Stats(C.Structure):
    _fields_ = [("lightVertices", C.c_longlong), ("gridVertices", C.c_longlong),
                ("lightRays", C.c_longlong), ("cameraRays", C.c_longlong),
                ("shadowRays", C.c_longlong), ("mergeQueries", C.c_longlong),
                ("mergeCandidates", C.c_longlong), ("mergeAccepted", C.c_longlong),
                ("connections", C.c_longlong), ("lightSplats", C.c_longlong),
                ("msLight", C.c_float), ("msGrid", C.c_float),
                ("msCamera", C.c_float), ("msTotal", C.c_float),
                ("msLightKernel", C.c_float), ("msCameraKernel", C.c_float),
                ("msMergeKernel", C.c_float), ("msQuerySort", C.c_float), ("msConnectKernels", C.c_float),
                ("radius", C.c_float)]

    def asdict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


# Scene::BoxMask (reference src/scene.hxx:112-126)
kLightCeiling, kLightSun, kLightPoint, kLightBackground = 1, 2, 4, 8
kLargeMirrorSphere, kLargeGlassSphere, kSmallMirrorSphere, kSmallGlassSphere = 16, 32, 64, 128
kGlossyFloor = 256
kBothSmallSpheres = kSmallMirrorSphere | kSmallGlassSphere

# g_SceneConfigs (reference src/config.hxx:146-151)
SCENE_CONFIGS = [
    kGlossyFloor | kBothSmallSpheres | kLightSun,
    kGlossyFloor | kLargeMirrorSphere | kLightCeiling,
    kGlossyFloor | kBothSmallSpheres | kLightPoint,
    kGlossyFloor | kBothSmallSpheres | kLightBackground,
]

"""Version-2 scenes from Python: a SceneBuilder that assembles a `vcm_scene_desc2` (any number of primitives,
materials and lights; include/smallvcm_amd.h) with the library's constructors -- vcm_make_triangle, vcm_make_area_light,
vcm_make_camera, ... each of which computes the derived members exactly as the reference's constructor does
(geometry.hxx:111-123, lights.hxx:116-127, camera.hxx:37-76, scene.hxx:387-398).  Host-only: needs no GPU.

    b = SceneBuilder()
    white = b.material(diffuse=(0.8, 0.8, 0.8))
    b.triangle(p0, p1, p2, white)
    b.emissive_triangle(q0, q1, q2, intensity=(25, 25, 25))
    scene = b.build(position, forward, up, fov_deg=45, resx=512, resy=512)
    r = VertexCM(scene, VertexCM.kVcm, 0.003, 0.75)          # more than 32 primitives: traced through a BVH
"""
import ctypes as C

from ._abi import Camera, Light, Material, Prim, SceneDesc2


def _f3(v):
    return (C.c_float * 3)(float(v[0]), float(v[1]), float(v[2]))


class SceneBuilder:
    def __init__(self):
        from .renderer import load_library
        self.L = load_library(require_gpu=False)
        L = self.L
        fp = C.POINTER(C.c_float)
        L.vcm_make_triangle.argtypes = [fp, fp, fp, C.c_int, C.POINTER(Prim)]
        L.vcm_make_triangle.restype = None
        L.vcm_make_sphere.argtypes = [fp, C.c_float, C.c_int, C.POINTER(Prim)]
        L.vcm_make_sphere.restype = None
        L.vcm_make_area_light.argtypes = [fp, fp, fp, fp, C.POINTER(Light)]
        L.vcm_make_area_light.restype = None
        L.vcm_make_directional_light.argtypes = [fp, fp, C.POINTER(Light)]
        L.vcm_make_directional_light.restype = None
        L.vcm_make_point_light.argtypes = [fp, fp, C.POINTER(Light)]
        L.vcm_make_point_light.restype = None
        L.vcm_make_background_light.argtypes = [C.c_float, C.POINTER(Light)]
        L.vcm_make_background_light.restype = None
        L.vcm_make_material.argtypes = [C.POINTER(Material)]
        L.vcm_make_material.restype = None
        L.vcm_make_camera.argtypes = [fp, fp, fp, C.c_float, C.c_int, C.c_int, C.POINTER(Camera)]
        L.vcm_make_scene_sphere.argtypes = [C.POINTER(Prim), C.c_int, fp, fp, fp]
        L.vcm_make_scene_sphere.restype = None
        self.prims, self.materials, self.mat2light, self.lights = [], [], [], []
        self.background = -1

    # ---- materials (materials.hxx:33-65) ----
    def material(self, diffuse=(0, 0, 0), phong=(0, 0, 0), exponent=1.0, mirror=(0, 0, 0), ior=-1.0):
        m = Material()
        self.L.vcm_make_material(C.byref(m))
        m.diffuse[:] = [float(x) for x in diffuse]
        m.phong[:] = [float(x) for x in phong]
        m.phongExp = float(exponent)
        m.mirror[:] = [float(x) for x in mirror]
        m.ior = float(ior)
        self.materials.append(m)
        self.mat2light.append(-1)
        return len(self.materials) - 1

    # ---- geometry ----
    def triangle(self, p0, p1, p2, material):
        p = Prim()
        self.L.vcm_make_triangle(_f3(p0), _f3(p1), _f3(p2), int(material), C.byref(p))
        self.prims.append(p)
        return len(self.prims) - 1

    def sphere(self, center, radius, material):
        p = Prim()
        self.L.vcm_make_sphere(_f3(center), float(radius), int(material), C.byref(p))
        self.prims.append(p)
        return len(self.prims) - 1

    # ---- lights ----
    def emissive_triangle(self, p0, p1, p2, intensity, diffuse=(0, 0, 0)):
        """a triangle that is an area light: its own material, whose mat2light entry names the light (scene.hxx:333-361)"""
        mat = self.material(diffuse=diffuse)
        light = Light()
        self.L.vcm_make_area_light(_f3(p0), _f3(p1), _f3(p2), _f3(intensity), C.byref(light))
        self.lights.append(light)
        self.mat2light[mat] = len(self.lights) - 1
        return self.triangle(p0, p1, p2, mat)

    def directional_light(self, direction, intensity):
        light = Light()
        self.L.vcm_make_directional_light(_f3(direction), _f3(intensity), C.byref(light))
        self.lights.append(light)

    def point_light(self, position, intensity):
        light = Light()
        self.L.vcm_make_point_light(_f3(position), _f3(intensity), C.byref(light))
        self.lights.append(light)

    def background_light(self, scale=1.0):
        light = Light()
        self.L.vcm_make_background_light(float(scale), C.byref(light))
        self.lights.append(light)
        self.background = len(self.lights) - 1

    # ---- the description ----
    def build(self, position, forward, up, fov_deg, resx, resy):
        d = SceneDesc2()
        prims = (Prim * max(len(self.prims), 1))(*self.prims)
        mats = (Material * len(self.materials))(*self.materials)
        m2l = (C.c_int * len(self.mat2light))(*self.mat2light)
        lights = (Light * len(self.lights))(*self.lights)
        d.nPrims, d.prims = len(self.prims), C.cast(prims, C.POINTER(Prim))
        d.nMaterials, d.materials, d.mat2light = len(self.materials), C.cast(mats, C.POINTER(Material)), C.cast(m2l, C.POINTER(C.c_int))
        d.nLights, d.lights = len(self.lights), C.cast(lights, C.POINTER(Light))
        d.backgroundLight = self.background
        r, inv = C.c_float(), C.c_float()
        self.L.vcm_make_scene_sphere(d.prims, d.nPrims, d.sceneCenter, C.byref(r), C.byref(inv))
        d.sceneRadius, d.invSceneRadiusSqr = r.value, inv.value
        if self.L.vcm_make_camera(_f3(position), _f3(forward), _f3(up), float(fov_deg), int(resx), int(resy), C.byref(d.camera)) != 0:
            raise ValueError("bad camera")
        d._keep = (prims, mats, m2l, lights)   # the arrays live as long as the description
        return d

#!/usr/bin/env python
"""How well do 64 lanes in lockstep use vcm_core.h's BVH traversal?  Runs the product's device functions on the host
(tests/host_emul, measurement build libemul_prof.so) over the mesh scene, logs every traversal's event stream and replays
it as waves of 64 rays:  lane cost / (64 x wave cost) = the lane utilisation the traversal would show on the device, for
primary rays, later bounces and shadow rays, as built and with dynamic fetch.   make -C tests/host_emul libemul_prof.so"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from mesh_scenes import bumpy_room  # noqa: E402
from smallvcm_amd._abi import SceneDesc2  # noqa: E402

E = C.CDLL(os.path.join(ROOT, "tests", "host_emul", "libemul_prof.so"))
E.emul_create2.restype = C.c_void_p
E.emul_create2.argtypes = [C.POINTER(SceneDesc2), C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int]
E.emul_run_iteration.argtypes = [C.c_void_p, C.c_int, C.c_uint, C.c_uint]
E.emul_bvh_log_size.restype = C.c_longlong
E.emul_bvh_simulate.argtypes = [C.c_int] * 4 + [C.c_double] * 4 + [C.POINTER(C.c_double)]

res = int(sys.argv[1]) if len(sys.argv) > 1 else 160
sc = bumpy_room(grid=72, resx=res, resy=res)
h = E.emul_create2(C.byref(sc), 4, 0.003, 0.75, 1234, 0, 1)
E.emul_run_iteration(h, 0, 0, 10)
print("%d primitives, %dx%d, one iteration: %d events" % (sc.nPrims, res, res, E.emul_bvh_log_size()))
CI, CP, CL, CT = 62.0, 36.0, 66.0, 450.0   # wave-instructions per inner step / pop with box test / leaf primitive / task set-up + evaluation


def sim(kind, which, path, refill=0, ct=0.0):
    out = (C.c_double * 9)()
    E.emul_bvh_simulate(ord(kind), which, ord(path) if path else 0, refill, CI, CP, CL, ct, out)
    return list(out)


print("%-46s %9s %7s %7s %6s %6s %9s %9s %7s" % ("rays", "count", "mean I", "mean L", "max I", "max L", "lane util", "wave/ray", "bound"))
for name, kind, which, path in [("camera: primary (bounce 1)", "B", 1, "c"), ("camera: bounce 2", "B", 2, "c"), ("camera: bounce 3", "B", 3, "c"),
                                ("camera: bounce 5", "B", 5, "c"), ("camera: all closest-hit rays, log order", "B", 0, "c"),
                                ("light: bounce 1", "B", 1, "l"), ("light: bounce 2", "B", 2, "l"), ("light: all closest-hit", "B", 0, "l"),
                                ("camera paths: shadow rays (DI + VC)", "b", 0, "c"), ("light paths: shadow rays (to the camera)", "b", 0, "l")]:
    o = sim(kind, which, path)
    print("%-46s %9d %7.1f %7.1f %6d %6d %9.3f %9.1f %7.3f" % (name, o[0], o[3], o[4], o[5], o[6], o[1] / max(o[2], 1), o[2] / 64.0 / max(o[0], 1), o[1] / max(o[8], 1)))
print("\ndynamic fetch (a finished lane takes the next ray once `refill` lanes are idle).  Traversal wave cost relative to one ray per lane,")
print("and the fetch rounds per 64 rays (each round runs the task's set-up / evaluation code for the lanes it serves; as built: 1):")
for name, kind, path in (("camera paths: shadow rays", "b", "c"), ("camera: closest-hit rays", "B", "c"), ("light: closest-hit rays", "B", "l")):
    base = sim(kind, 0, path)
    for refill in (48, 32, 24, 16, 8, 1):
        a, b = sim(kind, 0, path, refill, 0.0), sim(kind, 0, path, refill, 1.0)
        rounds = (b[2] - a[2]) / 64.0
        print("  %-26s refill %2d: traversal %.3f, lane util %.3f, fetch rounds per 64 rays %.2f" % (name, refill, a[2] / base[2], a[1] / a[2], rounds / (base[0] / 64.0)))

print("\nshadow rays ordered by a key before they are dealt to the waves -- what sorting the tasks of K3b / K3c would buy:")
E.emul_bvh_sort.argtypes = [C.c_int]
E.emul_bvh_sort_mode.argtypes = [C.c_int]
for mode, name in ((0, "origin cell, direction octant"), (1, "origin cell, end-point cell"), (3, "origin cell alone")):
    E.emul_bvh_sort_mode(mode)
    for cells in (0, 4, 8, 16, 32):
        E.emul_bvh_sort(cells)
        for kind, path, what in (("b", "c", "camera shadow rays"), ("b", "l", "light->camera rays")):
            o = sim(kind, 0, path)
            print("  %-30s grid %2d^3  %-20s lane util %.3f, wave cost per ray %.1f" % (name, cells, what, o[1] / max(o[2], 1), o[2] / 64.0 / max(o[0], 1)))
E.emul_bvh_sort(0)

"""Version-2 scenes (vcm_scene_desc2): any number of primitives, traced through a BVH.

not gpu
  * the library's constructors (vcm_make_*) build, bit for bit, what the REFERENCE's constructors build;
  * the oracle (brute force over every primitive, like the reference) equals the unmodified reference run on a Scene the
    harness assembles from the same arrays;
  * the product's device functions compiled for the host -- INCLUDING the BVH traversal -- equal the oracle: closest
    hit with the reference's tie rule, occlusion, whole iterations.
gpu: the same on the device, with a scene of more than 10 000 triangles."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib
from emul_lib import Emul, emul
from mesh_scenes import bumpy_room, tilted_room
from oracle_lib import Oracle
from smallvcm_amd._abi import Camera, Light, Prim, SceneDesc2

needs_ref = pytest.mark.skipif(not oracle_lib.have_ref(), reason="oracle/_ref not built (no /root/reference)")
_fp = C.POINTER(C.c_float)


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


@needs_ref
def test_constructors_equal_the_reference_constructors():
    from smallvcm_amd.renderer import load_library
    L, R = load_library(require_gpu=False), oracle_lib.ref_tape()
    for lib, pre in ((L, "vcm_make_"), (R, "ref_make_")):
        getattr(lib, pre + "triangle").argtypes = [_fp, _fp, _fp, C.c_int, C.POINTER(Prim)]
        getattr(lib, pre + "area_light").argtypes = [_fp, _fp, _fp, _fp, C.POINTER(Light)]
        getattr(lib, pre + "directional_light").argtypes = [_fp, _fp, C.POINTER(Light)]
        getattr(lib, pre + "camera").argtypes = [_fp, _fp, _fp, C.c_float, C.c_int, C.c_int, C.POINTER(Camera)]
    rng = np.random.default_rng(11)
    for _ in range(2000):
        p = (rng.random((3, 3)) * 4 - 2).astype(np.float32)
        a, b = Prim(), Prim()
        L.vcm_make_triangle(_f3(p[0]), _f3(p[1]), _f3(p[2]), 3, C.byref(a))
        R.ref_make_triangle(_f3(p[0]), _f3(p[1]), _f3(p[2]), 3, C.byref(b))
        assert bytes(a) == bytes(b)
        la, lb = Light(), Light()
        L.vcm_make_area_light(_f3(p[0]), _f3(p[1]), _f3(p[2]), _f3((1, 2, 3)), C.byref(la))
        R.ref_make_area_light(_f3(p[0]), _f3(p[1]), _f3(p[2]), _f3((1, 2, 3)), C.byref(lb))
        assert bytes(la) == bytes(lb)
        L.vcm_make_directional_light(_f3(p[0]), _f3((1, 2, 3)), C.byref(la))
        R.ref_make_directional_light(_f3(p[0]), _f3((1, 2, 3)), C.byref(lb))
        assert bytes(la) == bytes(lb)
    for _ in range(200):
        pos, fwd = (rng.random(3) * 6 - 3).astype(np.float32), (rng.random(3) * 2 - 1).astype(np.float32)
        up = np.float32([0.01, 0.02, 1.0])
        ca, cb = Camera(), Camera()
        fov, rx, ry = float(rng.random() * 60 + 20), int(rng.integers(16, 2048)), int(rng.integers(16, 2048))
        assert L.vcm_make_camera(_f3(pos), _f3(fwd), _f3(up), fov, rx, ry, C.byref(ca)) == 0
        R.ref_make_camera(_f3(pos), _f3(fwd), _f3(up), fov, rx, ry, C.byref(cb))
        assert bytes(ca) == bytes(cb)


@needs_ref
def test_a_built_scene_is_what_the_reference_would_hold():
    for kw in ({}, {"sun": True}, {"background": True, "spheres": False}):
        assert oracle_lib.ref_check_scene2(bumpy_room(grid=6, **kw)) == 0


def _kat_rays(n, seed):
    rng = np.random.default_rng(seed)
    a = np.zeros((n, 16), np.float32)
    a[:, 0:3] = (rng.random((n, 3)) * 2.3 - 1.15).astype(np.float32)
    d = rng.normal(size=(n, 3))
    a[:, 3:6] = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    k = n // 6   # grid-aligned origins and axis-parallel directions: shared edges and vertices of the floor mesh
    a[:k, 0:2] = np.round(a[:k, 0:2] * 12) / 12 * np.float32(1.25 / 1.0)
    a[k:2 * k, 3:6] = np.float32([0, 0, -1])
    return a


@needs_ref
@pytest.mark.parametrize("op,name", [(0, "intersect"), (1, "occluded")])
def test_bvh_traversal_on_the_host_equals_the_reference_walk(op, name):
    """Scene::Intersect / Occluded of the unmodified reference (every primitive, list order) against the product's BVH
    traversal (device functions compiled for the host): hit distance, normal, material, light -- bit for bit, the tie
    rule at shared edges included."""
    sc = bumpy_room(grid=20)
    assert sc.nPrims > 800
    rays = _kat_rays(30000, 3 + op)
    if op == 1:
        rays[:, 6] = (np.random.default_rng(9).random(len(rays)) * 3).astype(np.float32)
    want, mine = np.zeros_like(rays), np.zeros_like(rays)
    R, E = oracle_lib.ref_tape(), emul()
    R.ref_kat2.argtypes = [C.POINTER(SceneDesc2), C.c_int, C.c_int, _fp, _fp]
    E.emul_kat2.argtypes = [C.POINTER(SceneDesc2), C.c_int, C.c_int, _fp, _fp]
    assert R.ref_kat2(C.byref(sc), op, len(rays), rays.ctypes.data_as(_fp), want.ctypes.data_as(_fp)) == 0
    E.emul_kat2(C.byref(sc), op, len(rays), rays.ctypes.data_as(_fp), mine.ctypes.data_as(_fp))
    bad = np.nonzero((want.view(np.uint32) != mine.view(np.uint32)).any(axis=1))[0]
    assert len(bad) == 0, (name, len(bad), bad[:5])
    assert want[:, 0].sum() > 100


MESH_CASES = [({"grid": 10}, 4, 48, 2), ({"grid": 12, "sun": True}, 4, 40, 1), ({"grid": 8, "background": True, "spheres": False}, 4, 40, 1),
              ({"grid": 10}, 2, 40, 1), ({"grid": 10}, 3, 40, 1), ({"grid": 9}, 5, 48, 2), ({"grid": 10}, 0, 40, 1)]


@needs_ref
@pytest.mark.parametrize("kw,algo,res,nit", MESH_CASES)
def test_oracle_equals_reference_on_mesh_scenes(kw, algo, res, nit):
    sc = bumpy_room(resx=res, resy=res, **kw)
    o = Oracle(sc, algo, threads=8)
    lcs, ccs = [], []
    for it in range(nit):
        o.run_iteration(it, 0, 10)
        a, b = o.counts()
        lcs.append(a)
        ccs.append(b)
    fb, consumed, bad = oracle_lib.ref_run_tape2(sc, algo, np.concatenate(lcs), np.concatenate(ccs), n_iter=nit)
    assert bad == 0
    assert np.array_equal(fb.view(np.uint32), o.framebuffer().view(np.uint32))
    assert fb.max() > 0


@pytest.mark.parametrize("kw,algo,res,nit", MESH_CASES)
def test_device_functions_with_bvh_equal_oracle_on_mesh_scenes(kw, algo, res, nit):
    sc = bumpy_room(resx=res, resy=res, **kw)
    o, e = Oracle(sc, algo, threads=8), Emul(sc, algo)
    for it in range(nit):
        o.run_iteration(it, 0, 10)
        e.run_iteration(it, 0, 10)
    assert np.array_equal(o.framebuffer().view(np.uint32), e.framebuffer().view(np.uint32))
    for a, b in zip(o.counts(), e.counts()):
        assert np.array_equal(a, b)
    so, se = o.stats(), e.stats()
    for k in se:
        assert so[k] == se[k], k


TILTED_CASES = [({}, 4, 96, 2), ({"sun": True}, 4, 64, 1), ({}, 2, 64, 1), ({}, 3, 64, 1), ({"angle": 1.1}, 5, 64, 1),
                # Phong exponents other than the reference's 90 (integers that differ inside a wave; fractions: glibc's table walk)
                ({"exponents": (17.0, 200.0)}, 4, 48, 2), ({"exponents": (37.5, 90.0)}, 4, 48, 1), ({"exponents": (0.75, 12.25)}, 2, 40, 1)]


@needs_ref
@pytest.mark.parametrize("kw,algo,res,nit", TILTED_CASES)
def test_oracle_equals_reference_on_the_tilted_list_scene(kw, algo, res, nit):
    """pins the oracle on the scene the general filter path is tested with: the unmodified reference replaying the
    oracle's tape gives the same framebuffer, bit for bit"""
    sc = tilted_room(resx=res, resy=res, **kw)
    o = Oracle(sc, algo, threads=8)
    lcs, ccs = [], []
    for it in range(nit):
        o.run_iteration(it, 0, 10)
        a, b = o.counts()
        lcs.append(a)
        ccs.append(b)
    fb, consumed, bad = oracle_lib.ref_run_tape2(sc, algo, np.concatenate(lcs), np.concatenate(ccs), n_iter=nit)
    assert bad == 0
    assert np.array_equal(fb.view(np.uint32), o.framebuffer().view(np.uint32))
    assert fb.max() > 0


@pytest.mark.parametrize("kw,algo,res,nit", TILTED_CASES)
def test_device_functions_on_a_tilted_list_scene_equal_oracle(kw, algo, res, nit):
    """<= 32 primitives, none axis-aligned: the brute-force list with its GENERAL filter path (vcm_core.h: two plane
    parts per entry, single-triangle entries, pairs without a shared edge) against the oracle, on the host."""
    sc = tilted_room(resx=res, resy=res, **kw)
    assert sc.nPrims <= 32
    o, e = Oracle(sc, algo, threads=8), Emul(sc, algo)
    for it in range(nit):
        o.run_iteration(it, 0, 10)
        e.run_iteration(it, 0, 10)
    assert np.array_equal(o.framebuffer().view(np.uint32), e.framebuffer().view(np.uint32))
    for a, b in zip(o.counts(), e.counts()):
        assert np.array_equal(a, b)
    so, se = o.stats(), e.stats()
    for k in se:
        assert so[k] == se[k], k
    assert o.framebuffer().max() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("kw,algo,res,nit", [({}, 4, 256, 2), ({"sun": True}, 4, 128, 1), ({}, 2, 128, 1), ({"angle": 1.1}, 3, 128, 1), ({}, 5, 128, 2),
                                             # Phong exponents other than the reference's 90: two different integers in one
                                             # wave (the per-lane binary exponentiation), fractions (the general powf where a
                                             # lobe is evaluated: SceneList kernels, k_merge_walk<false>)
                                             ({"exponents": (17.0, 200.0)}, 4, 128, 2), ({"exponents": (37.5, 90.0)}, 4, 128, 2),
                                             ({"exponents": (0.75, 12.25)}, 2, 128, 1), ({"exponents": (37.5, 8.0)}, 5, 128, 1)])
def test_gpu_tilted_list_scene_equals_oracle_and_reference(kw, algo, res, nit):
    """The same on the GPU (SceneList kernels, general filter path): tape, counters, framebuffer bit for bit against the
    oracle and, through the tape, against the unmodified reference."""
    from smallvcm_amd.renderer import VertexCM
    sc = tilted_room(resx=res, resy=res, **kw)
    o = Oracle(sc, algo, threads=8)
    r = VertexCM(sc, algo, 0.003, 0.75, 1234)
    r.mMaxPathLength = 10
    lcs, ccs = [], []
    for it in range(nit):
        o.run_iteration(it, 0, 10)
        r.RunIteration(it)
        lc, cc = r.backend.rng_counts()
        olc, occ = o.counts()
        assert np.array_equal(lc, olc) and np.array_equal(cc, occ)
        lcs.append(lc)
        ccs.append(cc)
        so, sg = o.stats(), r.stats()
        for k in ("lightVertices", "lightRays", "cameraRays", "shadowRays", "mergeQueries", "mergeCandidates", "mergeAccepted",
                  "connections", "lightSplats"):
            assert so[k] == sg[k], (k, so[k], sg[k])
    fb = r.framebuffer_sum()
    r.close()
    assert np.array_equal(fb.view(np.uint32), o.framebuffer().view(np.uint32))
    assert fb.max() > 0
    if oracle_lib.have_ref() and res <= 128:
        rfb, consumed, bad = oracle_lib.ref_run_tape2(sc, algo, np.concatenate(lcs), np.concatenate(ccs), n_iter=nit)
        assert bad == 0
        assert np.array_equal(fb.view(np.uint32), rfb.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("grid,algo,res,nit,exponent", [(72, 4, 128, 2, 90.0), (72, 2, 96, 1, 90.0), (40, 5, 128, 2, 90.0), (72, 3, 96, 1, 90.0),
                                                        (40, 4, 96, 2, 37.5)])   # a fractional Phong exponent: the SceneBvhG kernels
def test_gpu_mesh_scene_through_the_bvh_equals_oracle_and_reference(grid, algo, res, nit, exponent):
    """> 10 000 triangles on the GPU (BVH) against the oracle's brute force -- tape, counters, framebuffer bit for bit --
    and, through the tape, against the unmodified reference."""
    from smallvcm_amd.renderer import VertexCM
    sc = bumpy_room(grid=grid, resx=res, resy=res, exponent=exponent)
    assert sc.nPrims > (10000 if grid >= 72 else 3000)
    import os
    o = Oracle(sc, algo, threads=os.cpu_count() or 1)
    r = VertexCM(sc, algo, 0.003, 0.75, 1234)
    r.mMaxPathLength = 10
    lcs, ccs = [], []
    for it in range(nit):
        o.run_iteration(it, 0, 10)
        r.RunIteration(it)
        lc, cc = r.backend.rng_counts()
        olc, occ = o.counts()
        assert np.array_equal(lc, olc) and np.array_equal(cc, occ)
        lcs.append(lc)
        ccs.append(cc)
        so, sg = o.stats(), r.stats()
        for k in ("lightVertices", "lightRays", "cameraRays", "shadowRays", "mergeQueries", "mergeCandidates", "mergeAccepted",
                  "connections", "lightSplats"):
            assert so[k] == sg[k], (k, so[k], sg[k])
    fb = r.framebuffer_sum()
    r.close()
    assert np.array_equal(fb.view(np.uint32), o.framebuffer().view(np.uint32))
    assert fb.max() > 0
    if oracle_lib.have_ref() and res <= 96:
        rfb, consumed, bad = oracle_lib.ref_run_tape2(sc, algo, np.concatenate(lcs), np.concatenate(ccs), n_iter=nit)
        assert bad == 0
        assert np.array_equal(fb.view(np.uint32), rfb.view(np.uint32))


@pytest.mark.gpu
def test_gpu_cornell_scenes_through_a_forced_bvh_equal_brute_force():
    """SMALLVCM_AMD_FORCE_BVH=1 sends the built-in boxes (spheres resting on the floor included) through the BVH: same
    image as the list walk."""
    import os
    import subprocess
    import sys
    code = ("import numpy as np, sys; sys.path.insert(0, %r); from smallvcm_amd.renderer import VertexCM, cornell_scene\n"
            "out = []\n"
            "for sid in range(4):\n"
            "    r = VertexCM(cornell_scene(sid, 96, 96), 4, 0.003, 0.75, 1234); r.mMaxPathLength = 10\n"
            "    r.RunIteration(0); r.RunIteration(1); out.append(r.framebuffer_sum()); r.close()\n"
            "np.save(sys.argv[1], np.stack(out))\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import tempfile
    d = tempfile.mkdtemp()
    imgs = []
    for force in ("0", "1"):
        path = os.path.join(d, "fb%s.npy" % force)
        env = dict(os.environ, SMALLVCM_AMD_FORCE_BVH=force)
        r = subprocess.run([sys.executable, "-c", code, path], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        imgs.append(np.load(path))
    assert np.array_equal(imgs[0].view(np.uint32), imgs[1].view(np.uint32))


# ---- scene files (include/smallvcm_amd.h: vcm_scene_load; smallvcm_amd/csrc/scene_file.cpp) ------------------------------
def _desc_arrays(d):
    import ctypes as C
    from smallvcm_amd._abi import Light, Material, Prim
    prims = np.frombuffer(C.string_at(d.prims, d.nPrims * C.sizeof(Prim)), np.uint32).reshape(d.nPrims, -1)
    mats = np.frombuffer(C.string_at(d.materials, d.nMaterials * C.sizeof(Material)), np.uint32).reshape(d.nMaterials, -1)
    lights = np.frombuffer(C.string_at(d.lights, d.nLights * C.sizeof(Light)), np.uint32).reshape(d.nLights, -1)
    m2l = np.array([d.mat2light[i] for i in range(d.nMaterials)])
    return prims, mats, lights, m2l


def test_scene_file_equals_the_procedural_scene(tmp_path):
    """OBJ + MTL + .vcmscene written from the procedural bumpy room load back as the same scene: every primitive's
    geometry, the material each primitive refers to, the lights behind the emissive triangles, the scene sphere and the
    camera, bit for bit (material NUMBERS differ: the loader numbers materials by first use); and the oracle renders the
    same image from both."""
    from mesh_scenes import write_bumpy_room_files
    from smallvcm_amd.scene_file import load_scene
    path = write_bumpy_room_files(str(tmp_path), grid=6)
    a, b = load_scene(path, 40, 32), bumpy_room(grid=6, resx=40, resy=32)
    pa, ma, la, m2la = _desc_arrays(a)
    pb, mb, lb, m2lb = _desc_arrays(b)
    assert a.nPrims == b.nPrims == 2 * 36 + 8 + 2 + 2 and a.nLights == b.nLights == 2
    assert np.array_equal(pa[:, 0], pb[:, 0]) and np.array_equal(pa[:, 2:], pb[:, 2:])      # type, p0, p1, p2, normal
    assert np.array_equal(ma[pa[:, 1]], mb[pb[:, 1]])                                        # the material OF each primitive
    assert np.array_equal(m2la[pa[:, 1]], m2lb[pb[:, 1]]) and np.array_equal(la, lb)         # ... and its light
    assert bytes(a.camera) == bytes(b.camera)
    assert (a.sceneRadius, list(a.sceneCenter), a.invSceneRadiusSqr) == (b.sceneRadius, list(b.sceneCenter), b.invSceneRadiusSqr)
    fbs = []
    for sc in (a, b):
        o = Oracle(sc, 4, threads=8)
        o.run_iteration(0, 0, 10)
        fbs.append(o.framebuffer())
    assert np.array_equal(fbs[0].view(np.uint32), fbs[1].view(np.uint32)) and fbs[0].max() > 0


def test_the_committed_scene_file_is_the_10k_triangle_room():
    """tests/scenes/bumpy_room.vcmscene (regenerate: python tests/mesh_scenes.py) = bumpy_room(grid=72): what bench.py
    renders as configuration M1"""
    import os
    from smallvcm_amd.scene_file import load_scene
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scenes", "bumpy_room.vcmscene")
    a, b = load_scene(path, 64, 64), bumpy_room(grid=72, resx=64, resy=64)
    pa, pb = _desc_arrays(a)[0], _desc_arrays(b)[0]
    assert a.nPrims == b.nPrims == 10380 and np.array_equal(pa[:, 2:], pb[:, 2:]) and bytes(a.camera) == bytes(b.camera)


def test_scene_file_errors_are_reported(tmp_path):
    from smallvcm_amd.scene_file import load_scene
    with pytest.raises(ValueError, match="cannot open"):
        load_scene(str(tmp_path / "nothing.vcmscene"), 8, 8)
    (tmp_path / "a.mtl").write_text("newmtl m\nKd 0.5 0.5 0.5\nnewmtl lamp\nKe 3 3 3\n")
    (tmp_path / "bad1.obj").write_text("mtllib a.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nusemtl nope\nf 1 2 3\n")
    with pytest.raises(ValueError, match="unknown material"):
        load_scene(str(tmp_path / "bad1.obj"), 8, 8)
    (tmp_path / "bad2.obj").write_text("mtllib a.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nusemtl m\nf 1 2 4\n")
    with pytest.raises(ValueError, match="out of range"):
        load_scene(str(tmp_path / "bad2.obj"), 8, 8)
    (tmp_path / "dark.obj").write_text("mtllib a.mtl\nv 0 0 0\nv 1 0 0\nv 0 1 0\nusemtl m\nf 1 2 3\n")
    with pytest.raises(ValueError, match="no light"):
        load_scene(str(tmp_path / "dark.obj"), 8, 8)
    # a bare .obj with an emissive material loads with the default camera; a polygon is fanned, relative indices work
    (tmp_path / "ok.obj").write_text("mtllib a.mtl\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nusemtl m\nf 1 2 3 4\nv 0 0 2\nv 1 0 2\nv 0 1 2\nusemtl lamp\nf -3 -2 -1\n")
    d = load_scene(str(tmp_path / "ok.obj"), 16, 16)
    assert d.nPrims == 3 and d.nLights == 1 and d.nMaterials == 2 and d.mat2light[1] == 0
    (tmp_path / "s.vcmscene").write_text("obj ok.obj\nfrobnicate 1 2 3\n")
    with pytest.raises(ValueError, match="unknown directive"):
        load_scene(str(tmp_path / "s.vcmscene"), 8, 8)
    # a line of any length is read whole (a 4096-byte buffer cut a large polygon in the middle of an index): a
    # 3000-gon on one `f` line of ~17 KB fans into 2998 triangles
    n = 3000
    import math
    verts = "".join("v %.6f %.6f 0\n" % (math.cos(2 * math.pi * i / n), math.sin(2 * math.pi * i / n)) for i in range(n))
    (tmp_path / "ngon.obj").write_text("mtllib a.mtl\n" + verts + "usemtl m\nf " + " ".join(str(i + 1) for i in range(n)) +
                                       "\nv 0 0 2\nv 1 0 2\nv 0 1 2\nusemtl lamp\nf -3 -2 -1\n")
    d = load_scene(str(tmp_path / "ngon.obj"), 16, 16)
    assert d.nPrims == n - 2 + 1
    # a material name defined twice (a second mtllib) is an error, not a silent duplicate
    (tmp_path / "b.mtl").write_text("newmtl m\nKd 0.1 0.1 0.1\n")
    (tmp_path / "twice.vcmscene").write_text("mtllib a.mtl\nmtllib b.mtl\nobj ok.obj\n")
    with pytest.raises(ValueError, match="defined twice"):
        load_scene(str(tmp_path / "twice.vcmscene"), 8, 8)


@pytest.mark.gpu
def test_gpu_scene_file_through_the_loader_equals_oracle():
    """the committed 10 380-primitive scene FILE, loaded by the library, traced through the BVH: tape, counters and
    framebuffer equal to the oracle's brute force over the same description, and the image equal to the procedural
    scene's (same geometry, other material numbers)"""
    import os
    from smallvcm_amd.renderer import VertexCM
    from smallvcm_amd.scene_file import load_scene
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scenes", "bumpy_room.vcmscene")
    sc = load_scene(path, 96, 96)
    o = Oracle(sc, 4, threads=os.cpu_count() or 1)
    r = VertexCM(sc, 4, 0.003, 0.75, 1234)
    r.mMaxPathLength = 10
    for it in range(2):
        o.run_iteration(it, 0, 10)
        r.RunIteration(it)
        lc, cc = r.backend.rng_counts()
        olc, occ = o.counts()
        assert np.array_equal(lc, olc) and np.array_equal(cc, occ)
        so, sg = o.stats(), r.stats()
        for k in ("lightVertices", "mergeCandidates", "mergeAccepted", "connections", "lightSplats", "shadowRays"):
            assert so[k] == sg[k], (k, so[k], sg[k])
    fb = r.framebuffer_sum()
    r.close()
    assert np.array_equal(fb.view(np.uint32), o.framebuffer().view(np.uint32)) and fb.max() > 0
    r2 = VertexCM(bumpy_room(grid=72, resx=96, resy=96), 4, 0.003, 0.75, 1234)
    r2.mMaxPathLength = 10
    r2.RunIteration(0)
    r2.RunIteration(1)
    assert np.array_equal(fb.view(np.uint32), r2.framebuffer_sum().view(np.uint32))
    r2.close()


def test_the_loader_survives_mangled_files(tmp_path):
    """vcm_scene_load is the one entry point that parses text a user wrote (smallvcm_amd/csrc/scene_file.cpp): whatever the
    files hold, it returns a scene or NULL with a message -- no crash, no exception across the C-ABI.  A small scene file is
    mangled 300 ways (bytes flipped, tokens replaced by junk / huge / negative numbers, lines cut, dropped or doubled) and
    loaded in a process of its own."""
    import os
    import subprocess
    import sys
    work = str(tmp_path)
    code = r'''
import ctypes as C, os, random, sys
sys.path.insert(0, %r)
from smallvcm_amd.renderer import load_library
L = load_library(require_gpu=False)
L.vcm_scene_load.restype = C.c_void_p
L.vcm_scene_load.argtypes = [C.c_char_p, C.c_int, C.c_int]
L.vcm_scene_file_free.argtypes = [C.c_void_p]
L.vcm_scene_load_error.restype = C.c_char_p
work = %r
base = {
 "s.vcmscene": "obj m.obj\nsphere 0.2 0.2 0.5 0.25 mirror\ncamera 0 -4 0.2 0 1 0 0 0 1 45\n",
 "m.obj": "mtllib m.mtl\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nusemtl white\nf 1 2 3 4\nv 0 0 2\nv 1 0 2\nv 0 1 2\nusemtl lamp\nf -3 -2 -1\nusemtl glossy\nf 1/1/1 2/2/2 3/3/3\n",
 "m.mtl": "newmtl white\nKd 0.8 0.8 0.8\nnewmtl glossy\nKd 0.1 0.1 0.1\nKs 0.7 0.7 0.7\nNs 90\nnewmtl mirror\nKs 1 1 1\nillum 3\nnewmtl lamp\nKe 25 25 25\n",
}
junk = ["", "-1", "0", "1e39", "-1e39", "nan", "inf", "99999999999999999999", "x", "/", "//", "1/", "#", "\x00", "-0", "3000000000", "f", "v", "usemtl", "\t"]
rng = random.Random(7)
loaded = refused = 0
for case in range(300):
    files = dict(base)
    name = rng.choice(list(files))
    text = files[name]
    how = rng.randrange(6)
    if how == 0:
        b = bytearray(text.encode())
        for _ in range(rng.randrange(1, 6)): b[rng.randrange(len(b))] = rng.randrange(256)
        data = bytes(b)
    else:
        lines = text.split("\n")
        i = rng.randrange(len(lines))
        toks = lines[i].split(" ")
        if how == 1 and toks: toks[rng.randrange(len(toks))] = rng.choice(junk)
        elif how == 2: toks = toks[:rng.randrange(len(toks) + 1)]
        elif how == 3: lines.insert(i, lines[i]); toks = lines[i].split(" ")
        elif how == 4: toks = []
        elif how == 5: toks = toks + [rng.choice(junk)] * rng.randrange(1, 40)
        lines[i] = " ".join(toks)
        data = "\n".join(lines).encode()
        if rng.random() < 0.2: data = data[:rng.randrange(len(data) + 1)]
    for n, t in files.items():
        open(os.path.join(work, n), "wb").write(data if n == name else t.encode())
    h = L.vcm_scene_load(os.path.join(work, "s.vcmscene").encode(), 16, 16)
    if h: loaded += 1; L.vcm_scene_file_free(h)
    else:
        refused += 1
        assert L.vcm_scene_load_error(), case
print("loaded", loaded, "refused", refused)
assert loaded > 20 and refused > 20
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), work)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.returncode, r.stdout[-300:], r.stderr[-600:])

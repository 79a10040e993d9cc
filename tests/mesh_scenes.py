"""Procedural version-2 test scenes (TEST INPUT, deterministic): a Cornell-like room whose floor is a tessellated
height field -- thousands of triangles, so the product traces it through its BVH while the oracle and the unmodified
reference walk every primitive (the reference has no acceleration structure, README:208-209)."""
import numpy as np

from smallvcm_amd.scene2 import SceneBuilder


def bumpy_room(grid=24, resx=64, resy=64, spheres=True, sun=False, background=False, seed=5, exponent=90.0):
    """2 * grid^2 floor triangles + 8 wall / ceiling triangles + 2 emissive triangles (+ 2 spheres)"""
    rng = np.random.default_rng(seed)
    b = SceneBuilder()
    white = b.material(diffuse=(0.803922, 0.803922, 0.803922))
    green = b.material(diffuse=(0.156863, 0.803922, 0.172549))
    red = b.material(diffuse=(0.803922, 0.152941, 0.152941))
    glossy = b.material(diffuse=(0.1, 0.1, 0.1), phong=(0.7, 0.7, 0.7), exponent=float(exponent))
    mirror = b.material(mirror=(1, 1, 1))
    glass = b.material(mirror=(1, 1, 1), ior=1.6)
    lo, hi = -1.25, 1.25
    # floor: height field z = -1.25 + bumps
    xs = np.linspace(lo, hi, grid + 1, dtype=np.float32)
    h = (0.04 * np.sin(3.1 * xs)[:, None] * np.cos(2.3 * xs)[None, :] + 0.01 * rng.random((grid + 1, grid + 1))).astype(np.float32)
    z = (np.float32(lo) + h).astype(np.float32)
    for i in range(grid):
        for j in range(grid):
            p00, p10 = (xs[i], xs[j], z[i, j]), (xs[i + 1], xs[j], z[i + 1, j])
            p01, p11 = (xs[i], xs[j + 1], z[i, j + 1]), (xs[i + 1], xs[j + 1], z[i + 1, j + 1])
            m = glossy if (i + j) % 3 else white
            b.triangle(p00, p10, p11, m)
            b.triangle(p11, p01, p00, m)
    c = [(lo, hi, lo), (hi, hi, lo), (hi, hi, hi), (lo, hi, hi), (lo, lo, lo), (hi, lo, lo), (hi, lo, hi), (lo, lo, hi)]
    b.triangle(c[0], c[1], c[2], white); b.triangle(c[2], c[3], c[0], white)       # back wall
    b.triangle(c[3], c[7], c[4], green); b.triangle(c[4], c[0], c[3], green)       # left
    b.triangle(c[1], c[5], c[6], red); b.triangle(c[6], c[2], c[1], red)           # right
    if not background:
        b.triangle(c[2], c[6], c[7], white); b.triangle(c[7], c[3], c[2], white)   # ceiling
    if spheres:
        b.sphere((-0.5, 0.3, -0.75), 0.4, mirror)
        b.sphere((0.55, -0.2, -0.8), 0.35, glass)
    if sun:
        b.directional_light((-1.0, 1.5, -1.0), (10.0, 4.0, 0.0))
    elif background:
        b.background_light(1.0)
    else:
        q = [(-0.25, -0.25, 1.2), (0.25, -0.25, 1.2), (0.25, 0.25, 1.2), (-0.25, 0.25, 1.2)]
        b.emissive_triangle(q[0], q[1], q[2], (25.0, 25.0, 25.0))
        b.emissive_triangle(q[2], q[3], q[0], (25.0, 25.0, 25.0))
    return b.build((-0.0439815, -4.12529, 0.222539), (0.00688625, 0.998505, -0.0542161), (3.73896e-4, 0.0542148, 0.998529), 45.0,
                   resx, resy)


def tilted_room(resx=64, resy=64, angle=0.37, sun=False, exponents=(90.0, 90.0)):
    """A Cornell-like room of at most 32 primitives whose every vertex is rotated about a skew axis: NO triangle pair is
    axis-aligned, so the brute-force list takes its general path (two plane parts per entry) instead of the one the
    reference's own boxes take; it also holds a lone triangle between two spheres (an entry with one triangle), two
    consecutive unrelated triangles (an entry whose triangles share no edge) and a sphere resting on the floor.
    `exponents`: Phong exponents of the floor and of the back wall's gloss -- other integers than the reference's 90, or
    fractions (the general powf of smallvcm_amd/csrc/detmath.h where a lobe is evaluated)."""
    axis = np.array([0.3, -0.5, 0.81], np.float64)
    axis /= np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    R = np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * (K @ K)

    def rot(p):
        return tuple(float(x) for x in np.float32(R @ np.array(p, np.float64)))
    b = SceneBuilder()
    white = b.material(diffuse=(0.803922, 0.803922, 0.803922))
    green = b.material(diffuse=(0.156863, 0.803922, 0.172549))
    red = b.material(diffuse=(0.803922, 0.152941, 0.152941))
    glossy = b.material(diffuse=(0.1, 0.1, 0.1), phong=(0.7, 0.7, 0.7), exponent=float(exponents[0]))
    gloss2 = white if exponents[1] == exponents[0] else b.material(diffuse=(0.5, 0.5, 0.5), phong=(0.3, 0.3, 0.3), exponent=float(exponents[1]))
    mirror = b.material(mirror=(1, 1, 1))
    glass = b.material(mirror=(1, 1, 1), ior=1.6)
    lo, hi = -1.25, 1.25
    c = [rot(v) for v in [(lo, hi, lo), (hi, hi, lo), (hi, hi, hi), (lo, hi, hi), (lo, lo, lo), (hi, lo, lo), (hi, lo, hi), (lo, lo, hi)]]
    b.triangle(c[0], c[4], c[5], glossy); b.triangle(c[5], c[1], c[0], glossy)     # floor
    b.triangle(c[0], c[1], c[2], gloss2); b.triangle(c[2], c[3], c[0], gloss2)     # back wall
    b.triangle(c[3], c[7], c[4], green); b.triangle(c[4], c[0], c[3], green)       # left
    b.triangle(c[1], c[5], c[6], red); b.triangle(c[6], c[2], c[1], red)           # right
    b.triangle(c[2], c[6], c[7], white); b.triangle(c[7], c[3], c[2], white)       # ceiling
    b.sphere(rot((-0.5, 0.3, lo + 0.4)), 0.4, mirror)                              # resting on the floor
    b.triangle(rot((0.1, 0.2, -0.3)), rot((0.7, 0.5, -0.2)), rot((0.3, 0.6, 0.4)), red)   # a lone triangle
    b.sphere(rot((0.55, -0.2, -0.8)), 0.35, glass)
    b.triangle(rot((-0.9, 0.8, -0.6)), rot((-0.4, 0.9, -0.5)), rot((-0.7, 0.7, 0.1)), green)   # two unrelated triangles
    b.triangle(rot((0.2, -0.6, 0.5)), rot((0.8, -0.4, 0.6)), rot((0.5, -0.7, 0.9)), white)
    if sun:
        b.directional_light(rot((-1.0, 1.5, -1.0)), (10.0, 4.0, 0.0))
    else:
        q = [rot(v) for v in [(-0.25, -0.25, 1.2), (0.25, -0.25, 1.2), (0.25, 0.25, 1.2), (-0.25, 0.25, 1.2)]]
        b.emissive_triangle(q[0], q[1], q[2], (25.0, 25.0, 25.0))
        b.emissive_triangle(q[2], q[3], q[0], (25.0, 25.0, 25.0))
    return b.build(rot((-0.0439815, -4.12529, 0.222539)), rot((0.00688625, 0.998505, -0.0542161)), rot((3.73896e-4, 0.0542148, 0.998529)),
                   45.0, resx, resy)


def write_bumpy_room_files(directory, grid=24, seed=5, name="bumpy_room"):
    """bumpy_room(grid, spheres=True) as scene FILES -- <name>.vcmscene, <name>.obj, <name>_light.obj, <name>.mtl (the
    format of smallvcm_amd/csrc/scene_file.cpp) -- with every number printed to nine significant digits, i.e. exactly.
    Same primitives in the same order as the procedural scene; -> path of the .vcmscene"""
    import os
    rng = np.random.default_rng(seed)
    lo, hi = -1.25, 1.25
    xs = np.linspace(lo, hi, grid + 1, dtype=np.float32)
    h = (0.04 * np.sin(3.1 * xs)[:, None] * np.cos(2.3 * xs)[None, :] + 0.01 * rng.random((grid + 1, grid + 1))).astype(np.float32)
    z = (np.float32(lo) + h).astype(np.float32)

    def f(v):
        return "%.9g" % float(np.float32(v))
    with open(os.path.join(directory, name + ".mtl"), "w") as m:
        m.write("# materials of the bumpy room (tests/mesh_scenes.py)\n")
        for nm, kd in (("white", (0.803922, 0.803922, 0.803922)), ("green", (0.156863, 0.803922, 0.172549)), ("red", (0.803922, 0.152941, 0.152941))):
            m.write("newmtl %s\nKd %s %s %s\n" % ((nm,) + tuple(f(x) for x in kd)))
        m.write("newmtl glossy\nKd %s %s %s\nKs %s %s %s\nNs 90\nillum 2\n" % (f(0.1), f(0.1), f(0.1), f(0.7), f(0.7), f(0.7)))
        m.write("newmtl mirror\nKs 1 1 1\nillum 3\n")
        m.write("newmtl glass\nKs 1 1 1\nNi %s\nillum 7\n" % f(1.6))
        m.write("newmtl lamp\nKe 25 25 25\n")
    with open(os.path.join(directory, name + ".obj"), "w") as o:
        o.write("# floor height field (%d x %d cells), walls, ceiling\nmtllib %s.mtl\n" % (grid, grid, name))
        for i in range(grid + 1):
            for j in range(grid + 1):
                o.write("v %s %s %s\n" % (f(xs[i]), f(xs[j]), f(z[i, j])))

        def vid(i, j):
            return i * (grid + 1) + j + 1
        current = None
        for i in range(grid):
            for j in range(grid):
                mat = "glossy" if (i + j) % 3 else "white"
                if mat != current:
                    o.write("usemtl %s\n" % mat)
                    current = mat
                o.write("f %d %d %d\nf %d %d %d\n" % (vid(i, j), vid(i + 1, j), vid(i + 1, j + 1), vid(i + 1, j + 1), vid(i, j + 1), vid(i, j)))
        c = [(lo, hi, lo), (hi, hi, lo), (hi, hi, hi), (lo, hi, hi), (lo, lo, lo), (hi, lo, lo), (hi, lo, hi), (lo, lo, hi)]
        for p in c:
            o.write("v %s %s %s\n" % tuple(f(x) for x in p))
        n0 = (grid + 1) * (grid + 1) + 1

        def quad(a, b, cc, d, mat):   # two triangles, the order of bumpy_room: (a, b, c), (c, d, a)
            o.write("usemtl %s\nf %d %d %d\nf %d %d %d\n" % (mat, n0 + a, n0 + b, n0 + cc, n0 + cc, n0 + d, n0 + a))
        quad(0, 1, 2, 3, "white")
        o.write("usemtl green\nf %d %d %d\nf %d %d %d\n" % (n0 + 3, n0 + 7, n0 + 4, n0 + 4, n0 + 0, n0 + 3))
        o.write("usemtl red\nf %d %d %d\nf %d %d %d\n" % (n0 + 1, n0 + 5, n0 + 6, n0 + 6, n0 + 2, n0 + 1))
        o.write("usemtl white\nf %d %d %d\nf %d %d %d\n" % (n0 + 2, n0 + 6, n0 + 7, n0 + 7, n0 + 3, n0 + 2))
    with open(os.path.join(directory, name + "_light.obj"), "w") as o:
        o.write("# the lamp: two emissive triangles under the ceiling (relative indices, a quad given as ONE polygon: fanned)\n")
        q = [(-0.25, -0.25, 1.2), (0.25, -0.25, 1.2), (0.25, 0.25, 1.2), (-0.25, 0.25, 1.2)]
        for p in q:
            o.write("v %s %s %s\n" % tuple(f(x) for x in p))
        # bumpy_room's second lamp triangle is (q2, q3, q0); a fan around q0 would give (q0, q2, q3): write both faces
        o.write("usemtl lamp\nf -4 -3 -2\nf -2/1 -1/1/1 -4//1\n")
    path = os.path.join(directory, name + ".vcmscene")
    with open(path, "w") as s:
        s.write("# the bumpy room of tests/mesh_scenes.py as a scene file (format: smallvcm_amd/csrc/scene_file.cpp)\n")
        s.write("obj %s.obj\n" % name)
        s.write("sphere %s %s %s %s mirror\n" % (f(-0.5), f(0.3), f(-0.75), f(0.4)))
        s.write("sphere %s %s %s %s glass\n" % (f(0.55), f(-0.2), f(-0.8), f(0.35)))
        s.write("obj %s_light.obj\n" % name)
        s.write("camera %s\n" % " ".join(f(x) for x in (-0.0439815, -4.12529, 0.222539, 0.00688625, 0.998505, -0.0542161,
                                                        3.73896e-4, 0.0542148, 0.998529, 45.0)))
    return path


if __name__ == "__main__":   # python tests/mesh_scenes.py: regenerate the committed scene file (10 380 primitives)
    import os
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scenes")
    os.makedirs(d, exist_ok=True)
    print(write_bumpy_room_files(d, grid=72))


def random_scene(seed, resx=48, resy=48):
    """A scene drawn from `seed` (TEST INPUT): a room of randomly displaced wall quads with random triangles and spheres
    inside, random materials (diffuse / Phong with integer or fractional exponents / mirror / glass and their mixes),
    and one of: area lights, a point light, a directional light, a background light, or a mix of two kinds.  Half the
    seeds stay at <= 32 primitives (the brute-force list), the others go through the BVH.  Nothing degenerate: every
    triangle has an area, every sphere a radius, every material some albedo."""
    rng = np.random.default_rng(1000 + seed)
    b = SceneBuilder()

    def u(lo, hi, n=None):
        return (rng.random(n) * (hi - lo) + lo).astype(np.float32) if n else np.float32(rng.random() * (hi - lo) + lo)

    mats = [b.material(diffuse=u(0.2, 0.8, 3))]
    for _ in range(int(rng.integers(3, 8))):
        kind = int(rng.integers(0, 6))
        expo = float(rng.choice([1.0, 2.0, 7.0, 90.0, 90.0, 500.0, 0.5, 3.25, 37.5, 1024.0, 70000.0]))
        if kind == 0:
            mats.append(b.material(diffuse=u(0.1, 0.9, 3)))
        elif kind == 1:
            mats.append(b.material(diffuse=u(0.0, 0.3, 3), phong=u(0.2, 0.65, 3), exponent=expo))
        elif kind == 2:
            mats.append(b.material(mirror=u(0.5, 1.0, 3)))
        elif kind == 3:
            mats.append(b.material(mirror=(1, 1, 1), ior=float(u(1.1, 2.2))))
        elif kind == 4:
            mats.append(b.material(diffuse=u(0.05, 0.3, 3), phong=u(0.05, 0.3, 3), exponent=expo, mirror=u(0.05, 0.3, 3)))
        else:
            mats.append(b.material(phong=u(0.3, 0.9, 3), exponent=expo))

    def mat():
        return mats[int(rng.integers(0, len(mats)))]

    def opaque():
        return mats[0]

    lo, hi = -1.25, 1.25
    c = np.array([(lo, hi, lo), (hi, hi, lo), (hi, hi, hi), (lo, hi, hi), (lo, lo, lo), (hi, lo, lo), (hi, lo, hi), (lo, lo, hi)], np.float32)
    c = c + u(-0.08, 0.08, c.size).reshape(c.shape)        # nothing axis-aligned
    light_kind = int(rng.integers(0, 6))                   # 0, 1 area; 2 point; 3 directional; 4 background; 5 area + point
    walls = [(0, 1, 2, 3), (3, 7, 4, 0), (1, 5, 6, 2), (0, 4, 5, 1)]   # back, left, right, floor
    if light_kind not in (3, 4):
        walls.append((2, 6, 7, 3))                          # a ceiling unless the light comes from outside
    for q in walls:
        m = mat() if rng.random() < 0.5 else opaque()
        b.triangle(c[q[0]], c[q[1]], c[q[2]], m)
        b.triangle(c[q[2]], c[q[3]], c[q[0]], m)
    big = seed % 2 == 1
    for _ in range(int(rng.integers(40, 400)) if big else int(rng.integers(0, 8))):
        p = u(-1.0, 1.0, 3)
        size = float(u(0.05, 0.5))
        e1, e2 = u(-1, 1, 3) * np.float32(size), u(-1, 1, 3) * np.float32(size)
        if np.linalg.norm(np.cross(e1, e2)) < 1e-3:
            continue
        b.triangle(p, p + e1, p + e2, mat())
    for _ in range(int(rng.integers(0, 5))):
        b.sphere(u(-0.9, 0.9, 3), float(u(0.08, 0.45)), mat())
    if light_kind in (0, 1, 5):
        for _ in range(1 + light_kind % 2):
            p = np.float32([u(-0.6, 0.6), u(-0.6, 0.6), u(0.9, 1.15)])
            e1, e2 = np.float32([u(0.2, 0.5), u(-0.1, 0.1), u(-0.05, 0.05)]), np.float32([u(-0.1, 0.1), u(0.2, 0.5), u(-0.05, 0.05)])
            b.emissive_triangle(p, p + e1, p + e2, u(5.0, 40.0, 3))     # faces down: e1 x e2 points up, the reference's light emits along -normal? both signs occur
            if rng.random() < 0.5:
                b.emissive_triangle(p + e1 + e2, p + e2, p + e1, u(5.0, 40.0, 3))
    if light_kind in (2, 5):
        b.point_light(u(-0.7, 0.7, 3), u(10.0, 70.0, 3))
    if light_kind == 3:
        b.directional_light((float(u(-1, 1)), float(u(-1, 1)), -1.0), u(1.0, 12.0, 3))
    if light_kind == 4:
        b.background_light(float(u(0.5, 2.0)))
    return b.build((float(u(-0.3, 0.3)), -4.1, float(u(-0.2, 0.4))), (float(u(-0.05, 0.05)), 1.0, float(u(-0.08, 0.02))),
                   (3.7e-4, 0.054, 0.9985), float(u(35, 60)), resx, resy)

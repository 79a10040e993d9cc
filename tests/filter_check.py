#!/usr/bin/env python
"""Self-check of the certified filters (vcm_core.h) on the CPU: TEST infrastructure, not a pytest (minutes).

    python tests/filter_check.py [resolution=160] [iterations=3]

Builds the host emulation with -DVCM_FILTER_CHECK (tests/host_emul/libemul_check.so): scene_intersect /
scene_occluded then run the filter AND the reference's brute-force loop on every ray and print a line whenever the
filter called an answer certain that the loop does not give.  Drives VCM over the reference's four scene
configurations (rectangle filter), the same with SMALLVCM_AMD_NO_RECTS=1 (Pluecker filter, one plane per quad) and the
tilted room (general list), path lengths 10 and 24, and reports rays, fall-back rates and mismatches."""
import ctypes as C
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)


def build():
    out = os.path.join(HERE, "host_emul", "libemul_check.so")
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-ffp-contract=off", "-shared", "-DVCM_FILTER_CHECK", "-include", "cstdio",
           "-o", out, os.path.join(HERE, "host_emul", "emul.cpp"), os.path.join(ROOT, "smallvcm_amd", "csrc", "scene_cornell.cpp")]
    subprocess.run(cmd, check=True, cwd=os.path.join(HERE, "host_emul"))
    return out


def main():
    res = int(sys.argv[1]) if len(sys.argv) > 1 else 160
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    lib = build()
    import emul_lib
    emul_lib._E = None
    real = C.CDLL

    def cdll(path, *a, **k):   # emul_lib loads host_emul/libemul.so: hand it the checking build instead
        return real(lib if str(path).endswith("libemul.so") else path, *a, **k)
    C.CDLL = cdll
    E = emul_lib.emul()
    C.CDLL = real
    E.emul_filter_stats.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
    from mesh_scenes import tilted_room
    total = [0, 0, 0, 0]

    def run(name, scene):
        for max_len in (10, 24):
            e = emul_lib.Emul(scene, 4)
            for it in range(iters):
                e.run_iteration(it, 0, max_len)
            st = (C.c_ulonglong * 4)()
            E.emul_filter_stats(st, 1)
            for k in range(4):
                total[k] += st[k]
            print("%-26s maxLen %2d: %9d closest-hit rays (%.2e fall back), %9d shadow rays (%.2e fall back)"
                  % (name, max_len, st[0], st[1] / max(st[0], 1), st[2], st[3] / max(st[2], 1)), flush=True)

    for rects in (True, False):
        if rects:
            os.environ.pop("SMALLVCM_AMD_NO_RECTS", None)
        else:
            os.environ["SMALLVCM_AMD_NO_RECTS"] = "1"
        from smallvcm_amd.renderer import cornell_scene   # the library only builds the description (no GPU needed)
        for sid in range(4):
            run("scene %d %s" % (sid, "rectangles" if rects else "quads (Pluecker)"), cornell_scene(sid, res, res))
    os.environ.pop("SMALLVCM_AMD_NO_RECTS", None)
    run("tilted room (general list)", tilted_room(resx=res, resy=res))
    print("total: %d closest-hit + %d shadow rays; a 'FILTER MISMATCH' line above = a certified answer that differs from the "
          "reference's loop (none expected)" % (total[0], total[2]))


if __name__ == "__main__":
    main()

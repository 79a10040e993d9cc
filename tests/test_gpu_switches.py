"""Every environment switch of the library that selects another launch shape, stream layout, kernel or filter is a
shipped code path in front of RangeQuery::Process / Scene::Intersect (src/vertexcm.hxx:130-169, src/hashgrid.hxx:110-169,
src/scene.hxx:53-85): each must render the frames of the default build, bit for bit, and count the same work.  The
switches are read once per process, so every case is a process of its own -- the C++ host (no Python, no torch).  GPU only."""
import json
import os
import subprocess

import numpy as np
import pytest

from oracle_lib import Oracle
from smallvcm_amd.renderer import cornell_scene

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "smallvcm_amd", "host", "vcm_render")

# (scene, algorithm name, oracle algorithm id, resolution, iterations)
CONFIGS = [(1, "vcm", 4, 256, 2), (3, "bpm", 2, 192, 2)]
SWITCHES = [
    "SMALLVCM_AMD_MERGE=walk",                 # k_merge_walk (the default is k_merge_pairs)
    "SMALLVCM_AMD_NO_RECTS=1",                 # the Pluecker filter (SceneQuads kernels) instead of the rectangles
    "SMALLVCM_AMD_NO_RECTS=1 SMALLVCM_AMD_NO_ONEPLANE=1",   # ... and the general list (SceneList kernels)
    "SMALLVCM_AMD_FORCE_BVH=1",                # the boxes through the BVH kernels
    "SMALLVCM_AMD_GENERAL_POW=1",              # the kernels that keep the general powf (non-integer Phong exponents)
    "SMALLVCM_AMD_GENERAL_POW=1 SMALLVCM_AMD_FORCE_BVH=1",   # SceneBvhG
    "SMALLVCM_AMD_SPLAT_LONG=8",
    "SMALLVCM_AMD_TIMING=events",
    "SMALLVCM_AMD_STRICT_ORDER=1 SMALLVCM_AMD_ARENAS=1",
    "SMALLVCM_AMD_GRID_SORT=count",            # HashGrid::Build as rounds 1-5 did it: a counter per cell, one atomic per vertex, a ranking pass
    # launch shapes, all through the ONE switch (round 6; rounds 2-5: ten variables)
    "SMALLVCM_AMD_SHAPE=trace_waves=512,trace_chunk=64,light_waves=1024,task_blocks=96",
    "SMALLVCM_AMD_SHAPE=merge_blocks=64,merge_chunk=3,aux_blocks=64,resolve_blocks=3,buckets_per_path=1",
    "SMALLVCM_AMD_SHAPE=merge_blocks=16384,task_blocks=3072,trace_waves=4096,buckets_per_path=16",   # the launch shapes of a 2048^2 frame
    "SMALLVCM_AMD_SHAPE=grid_sort_blocks=1",   # the radix sort with ONE workgroup: ~70 000 vertices = 35 tiles in a row
    "SMALLVCM_AMD_SHAPE=grid_sort_blocks=7",   # chunks that are not a multiple of the tile
    "SMALLVCM_AMD_SHAPE=grid_sort_blocks=4096",   # more workgroups than 256-vertex chunks: most of them have nothing to do
    "SMALLVCM_AMD_GRID_SORT=count SMALLVCM_AMD_SHAPE=resolve_blocks=2048,merge_blocks=64",
    "SMALLVCM_AMD_MERGE=walk SMALLVCM_AMD_SHAPE=merge_blocks=64,merge_chunk=3",
]

_oracle_cache = {}


def _oracle_frame(cfg):
    if cfg not in _oracle_cache:
        sid, _, algo, res, iters = cfg
        o = Oracle(cornell_scene(sid, res, res), algo, threads=8)
        for it in range(iters):
            o.run_iteration(it, 0, 10)
        _oracle_cache[cfg] = (o.framebuffer() * np.float32(1.0 / iters), o.stats())
    return _oracle_cache[cfg]


def _read_pfm(path):
    raw = open(path, "rb").read()
    a, b, c, data = raw.split(b"\n", 3)
    w, h = (int(x) for x in b.split())
    assert a == b"PF" and c == b"-1"
    return np.frombuffer(data, np.float32).reshape(h, w, 3)


@pytest.mark.skipif(not os.path.exists(HOST), reason="vcm_render not built")
@pytest.mark.parametrize("switch", ["(default)"] + SWITCHES)
def test_switch_renders_the_default_frames(tmp_path, switch):
    env = dict(os.environ)
    strict = False
    if switch != "(default)":
        for kv in switch.split():
            k, v = kv.split("=", 1)
            env[k] = v
            strict = strict or k == "SMALLVCM_AMD_STRICT_ORDER"
    for cfg in CONFIGS:
        sid, name, algo, res, iters = cfg
        out = str(tmp_path / ("s%d.pfm" % sid))
        r = subprocess.run([HOST, "-s", str(sid), "-a", name, "-i", str(iters), "--res", str(res), str(res), "-o", out, "--json"],
                           capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, (switch, r.stdout[-1000:] + r.stderr[-2000:])
        info = json.loads(r.stdout.strip().splitlines()[-1])
        ref, stats = _oracle_frame(cfg)
        img = _read_pfm(out)
        if strict and algo == 4:   # strict mode splats with fp32 atomics (DESIGN "Two modes")
            assert np.allclose(img, ref, rtol=2e-5, atol=2e-7), switch
        else:
            assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), (switch, cfg)
        for k in ("lightVertices", "lightRays", "cameraRays", "shadowRays", "mergeQueries", "mergeCandidates", "mergeAccepted",
                  "connections", "lightSplats"):
            assert info["last_iteration_counters"][k] == stats[k], (switch, cfg, k)
